"""dev: where does the gated overlap (S2VT_OVERLAP=2) pay?  REINFORCE and XE steps at several row counts N = K * B <= 256 (the one-part persistent
backward recurrences at 64 / 128 / 256 rows), each setting in a child process, step time from HIP events.  env: STEPS (default 60)."""
import json, os, subprocess, sys
CODE = r'''
import os, sys, json
sys.path.insert(0, os.environ["S2VT_ROOT"])
import numpy as np, torch
import s2vt_amd
from s2vt_amd import model as M, hostglue, ops
B, K, steps = int(os.environ["AB_B"]), int(os.environ["AB_K"]), int(os.environ.get("STEPS", "60"))
V, H, E, Tc = 12000, 1000, 500, 20
mdl = M.Video_Caption_Generator(1536, V, E, H, B, 0, 5, Tc, seed=5, multisample=max(K, 1))
g = torch.Generator().manual_seed(1)
video = (torch.randn(B, 5, 1536, generator=g) * 0.5).abs().cuda()
rng = np.random.default_rng(B)
cap = rng.integers(2, V, (B, Tc)).astype(np.int32); cap[:, -1] = 0
gt = torch.as_tensor(cap).cuda(); gm = torch.as_tensor(hostglue.masks_from_ids(cap)).cuda()
r = (torch.rand(max(K, 1) * B, generator=g) * 2).cuda(); b = (torch.rand(B, generator=g) * 2).repeat(max(K, 1)).cuda()
def step(i):
    if K == 0:
        return mdl.xe_update(video, gt, gm, lr=1e-4, q1=True)
    s, _ = mdl.sample(video, K, True, seed=7 + i)
    return mdl.reinforce_update(video, s, None, r, b, lr=1e-6, reuse_sampler_state=True)
for i in range(8): step(i)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(steps): step(8 + i)
e1.record(); torch.cuda.synchronize()
print(json.dumps({"B": B, "K": K, "rows": max(K, 1) * B, "overlap": os.environ.get("S2VT_OVERLAP"), "ms_per_step": round(e0.elapsed_time(e1) / steps, 4), "timeouts": ops.chain_timeouts()}))
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for (B, K) in ((64, 0), (128, 0), (256, 0), (32, 2), (32, 4), (64, 2), (64, 4), (32, 8)):
    for rep in range(2):
        for mode in ("0", "2"):
            r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, S2VT_ROOT=root, S2VT_OVERLAP=mode, AB_B=str(B), AB_K=str(K)), capture_output=True, text=True, timeout=900)
            print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-1500:], flush=True)
