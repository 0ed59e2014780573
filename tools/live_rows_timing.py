"""Dev helper: the REINFORCE update (teacher-forced forward, losses, backward, clip, Adam -- no sampler) at the bench dimensions on
captions of MSVD-like lengths (1 + min(Poisson(6), Tc - 2) words + <eos>, what a trained captioner's samples look like): the
dense pass, the truncated unroll, and truncated unroll + live rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import s2vt_amd
from s2vt_amd import model as M

B, K, TC, V = 64, 5, 20, 12000
mdl = M.Video_Caption_Generator(1536, V, 500, 1000, B, 0, 5, TC, seed=1234)
g = torch.Generator().manual_seed(1)
video = (torch.randn(B, 5, 1536, generator=g) * 0.5).abs().cuda()
rng = np.random.default_rng(0)
N = K * B
ln = 1 + np.minimum(rng.poisson(6, N), TC - 2)
cap = rng.integers(2, V, (N, TC)).astype(np.int32)
for n in range(N):
    cap[n, ln[n]:] = 0
mask = (np.arange(TC)[None, :] <= ln[:, None]).astype(np.float32)
capd = torch.as_tensor(cap).cuda()
r = (rng.random(N) * 2).astype(np.float32); b = np.tile((rng.random(B) * 2).astype(np.float32), K)
print(f"live positions {int(mask.sum())} of {N * TC} ({mask.mean():.2f}), longest caption {mdl.active_steps(mask)} of {TC} steps")


def timeit(kw, n=30):
    for _ in range(5):
        mdl.reinforce_update(video, capd, mask, r, b, lr=1e-6, **kw)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        mdl.reinforce_update(video, capd, mask, r, b, lr=1e-6, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, kw in (("dense", dict(active_steps=None, live_mask=None)), ("truncated unroll", dict(active_steps="auto", live_mask=None)),
                 ("truncated unroll + live rows", dict(active_steps="auto", live_mask="auto"))):
    print(f"{name:<32} {timeit(kw):.3f} ms per update")
