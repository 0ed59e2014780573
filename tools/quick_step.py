"""Dev helper: time one REINFORCE step at the BASELINE config (not the bench contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import s2vt_amd
from s2vt_amd import model as M

B, K, Tc = 64, 5, 20
mdl = M.Video_Caption_Generator(1536, 12000, 500, 1000, B, 0, 5, Tc)
g = torch.Generator().manual_seed(1234)
video = (torch.randn(B, 5, 1536, generator=g) * 0.5).abs().cuda()
r = (torch.rand(K * B, generator=g) * 2).cuda(); b = (torch.rand(B, generator=g) * 2).repeat(K).cuda()


def step(i):
    s, gr = mdl.sample(video, K, True, seed=1000 + i)
    is_eos = (s == 0)
    mask = ((torch.cumsum(is_eos.int(), 1) - is_eos.int()) == 0).float()
    return mdl.reinforce_update(video, s, mask, r, b, lr=1e-6, reuse_sampler_state=True)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for i in range(2):
    step(i)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
t0 = time.time()
ev[0].record(); s, gr = mdl.sample(video, K, True, seed=5); ev[1].record()
torch.cuda.synchronize(); print("sample ms", ev[0].elapsed_time(ev[1]))
t0 = time.time()
for i in range(n):
    st = step(i + 2)
torch.cuda.synchronize()
dt = (time.time() - t0) / n
print(f"step {dt*1e3:.2f} ms  -> {K*B*Tc/dt:.0f} sampled tokens/s; loss {float(st.loss):.5f}")
