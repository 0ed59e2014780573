"""Dev helper: time the weight-gradient (TN) contractions of one REINFORCE step through bptt timing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import s2vt_amd
from s2vt_amd import model as M, ops
B, K, Tc = 64, 5, 20
mdl = M.Video_Caption_Generator(1536, 12000, 500, 1000, B, 0, 5, Tc)
g = torch.Generator().manual_seed(1234)
video = (torch.randn(B, 5, 1536, generator=g) * 0.5).abs().cuda()
r = (torch.rand(K * B, generator=g) * 2).cuda(); b = (torch.rand(B, generator=g) * 2).repeat(K).cuda()
s, gr = mdl.sample(video, K, True, seed=5)
mask = torch.ones_like(s, dtype=torch.float32)
for i in range(2):
    mdl.reinforce_update(video, s, mask, r, b, lr=1e-6)
torch.cuda.synchronize()
ops.prof_enable(True)
for i in range(5):
    mdl.reinforce_update(video, s, mask, r, b, lr=1e-6)
torch.cuda.synchronize()
ops.prof_enable(False)
for row in sorted(ops.prof_collect(), key=lambda r: -r["total_ms"]):
    print(f"class {row['kernel_class']} {row['name']:<20} launches {row['launches']:5d}  ms/step {row['total_ms']/5:7.3f}  {row['total_flops']/row['total_ms']/1e9:6.1f} TF")
