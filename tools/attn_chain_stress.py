"""dev: race hunt for the persistent attention recurrences -- the same step many times; the forward must reproduce its saved
activations bit for bit every time (and equal the per-step launches), the backward its gradients to the order-free noise."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import s2vt_amd
from s2vt_amd import attention as A, ops

N = int(os.environ.get("N", "300"))
D, V, H, Tv, Tc, B = 1536, 12000, 1000, int(os.environ.get("TV", "5")), 20, int(os.environ.get("B", "64"))
m = A.Attention_Caption_Generator(D, V, H, B, Tv, Tc, 0.9)
rng = np.random.default_rng(0)
video = torch.as_tensor(np.abs(rng.standard_normal((B, Tv, D)) * 0.5).astype(np.float32)).cuda()
cap = rng.integers(1, V, (B, Tc)).astype(np.int32)
mask = np.ones((B, Tc), np.float32)
capd = torch.as_tensor(cap).cuda()
vid, sid = m._row_ids(B)
with ops.chain_hold():
    ref, _, _ = ops.attn_teacher_forced_fwd(m.dims, m.store.params, video, capd, 0.9, 77, vid, sid)
    ref = ref.clone()
    m.xe_update(video, cap, mask, lr=0.0, active_steps=None); m.global_step = 0
    gref = m.store.grad.clone()
bad_f = bad_b = 0
worst = 0.0
for i in range(N):
    lg, _, _ = ops.attn_teacher_forced_fwd(m.dims, m.store.params, video, capd, 0.9, 77, vid, sid)
    if not torch.equal(lg, ref):
        bad_f += 1
    m.xe_update(video, cap, mask, lr=0.0, active_steps=None); m.global_step = 0
    err = float((m.store.grad - gref).abs().max() / gref.abs().max())
    worst = max(worst, err)
    if err > 1e-4:
        bad_b += 1
torch.cuda.synchronize()
print(f"{N} iterations: forward mismatches {bad_f}, backward outliers {bad_b} (worst relative deviation {worst:.2e}), timeouts {ops.chain_timeouts()}")
