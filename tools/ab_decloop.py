"""Dev helper: the sampler at the bench dimensions, per-step launches vs the persistent decode loop (S2VT_DECLOOP=1 in a child)."""
import os, subprocess, sys
CODE = r'''
import os, sys
sys.path.insert(0, os.environ["S2VT_ROOT"])
import torch, numpy as np
import s2vt_amd
from s2vt_amd import model as M
B, K = int(os.environ.get("DL_B", "64")), int(os.environ.get("DL_K", "5"))
mdl = M.Video_Caption_Generator(1536, 12000, 500, 1000, B, 0, 5, 20, seed=1234)
video = (torch.randn(B, 5, 1536, generator=torch.Generator().manual_seed(1)) * 0.5).abs().cuda()
for _ in range(5): s, g = mdl.sample(video, K, True, seed=3)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(30): s, g = mdl.sample(video, K, True, seed=3 + i)
e1.record(); torch.cuda.synchronize()
print("DECLOOP", os.environ.get("S2VT_DECLOOP"), "sample() ms", e0.elapsed_time(e1) / 30, "timeouts", s2vt_amd.ops.chain_timeouts(), "ids", int(s.sum()))
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for flag in ("0", "1"):
    r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, S2VT_ROOT=root, S2VT_DECLOOP=flag), capture_output=True, text=True, timeout=600)
    print(r.stdout.strip() or r.stderr[-2000:])
