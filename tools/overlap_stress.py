"""dev: race hunt for the gated overlap (csrc/train.hip, S2VT_OVERLAP=2): the same XE / mixed update (lr = 0) many times back to back -- every
repetition's gradients must equal the first one's up to the order-free reductions' noise, no grid-wide wait may time out, and the step time must not
show the cliff of a persistent grid that found the chip occupied (a contraction of the previous call still running).  env: ITERS (default 300)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import s2vt_amd
from s2vt_amd import model as M, hostglue, ops

ITERS = int(os.environ.get("ITERS", "300"))
for (B, K, name) in ((64, 0, "xe"), (32, 1, "multitask")):
    V, H, E, Tc = 12000, 1000, 500, 20
    mdl = M.Video_Caption_Generator(1536, V, E, H, B, 0, 5, Tc, seed=5, multisample=max(K, 1), label_dim=400 if K else 0, alpha=0.05 if K else 0.0)
    rng = np.random.default_rng(B)
    video = torch.as_tensor(np.abs(rng.standard_normal((B, 5, 1536)) * 0.5).astype(np.float32)).cuda()
    ln = 1 + np.minimum(rng.poisson(6, B), Tc - 2)
    cap = rng.integers(2, V, (B, Tc)).astype(np.int32)
    for j in range(B):
        cap[j, ln[j]:] = 0
    mask = hostglue.masks_from_ids(cap)
    labels = torch.as_tensor((rng.random((B, 400)) < 0.02).astype(np.float32)).cuda()
    r = torch.as_tensor(rng.random(max(K, 1) * B).astype(np.float32)).cuda(); b = torch.as_tensor(rng.random(B).astype(np.float32)).cuda()
    if K:
        s, _ = mdl.sample(video, K, True, seed=3)
        sm = hostglue.masks_from_ids(s.cpu().numpy())

    def step():
        mdl.global_step = 0                                    # the same dropout masks every time
        if K == 0:
            return mdl.xe_update(video, cap, mask, lr=0.0, q1=True)
        return mdl.mixed_update(video, s, sm, r, b, cap, mask, lr=0.0, lambda_loss=0.5, true_labels=labels, decay_all=True)
    step(); torch.cuda.synchronize()
    g0 = mdl.store.grad[:mdl.store.numel].clone()
    scale = float(g0.abs().max())
    worst, times = 0.0, []
    for i in range(ITERS):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); step(); e1.record(); torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
        worst = max(worst, float((mdl.store.grad[:mdl.store.numel] - g0).abs().max()) / scale)
    t = np.sort(np.asarray(times))
    print(f"{name}: {ITERS} repetitions, worst gradient deviation {worst:.2e} of the largest entry, step ms median {t[len(t)//2]:.3f} p99 {t[int(0.99*len(t))]:.3f} max {t[-1]:.3f}, "
          f"timeouts {ops.chain_timeouts()}, overlap mode {os.environ.get('S2VT_OVERLAP', '2 (default)')}")
    assert worst <= 5e-5 and ops.chain_timeouts() == 0
    del mdl
    torch.cuda.empty_cache()
