"""Dev helper: one REINFORCE step at the REFERENCE's own default configuration (batch 256, 8 samples, Tc = 35, |V| = 9972:
reinforcement_multisampling_tf_s2vt.py:505-517) -- 2.9 GB of logits per pass, operands beyond 2 GiB."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import s2vt_amd
from s2vt_amd import model as M

B, K, Tc, V = 256, 8, 35, 9972
mdl = M.Video_Caption_Generator(1536, V, 500, 1000, B, 0, 5, Tc)
g = torch.Generator().manual_seed(1234)
video = (torch.randn(B, 5, 1536, generator=g) * 0.5).abs().cuda()
r = (torch.rand(K * B, generator=g) * 2).cuda(); b = (torch.rand(B, generator=g) * 2).repeat(K).cuda()


def step(i):
    s, gr = mdl.sample(video, K, True, seed=1000 + i)
    is_eos = (s == 0)
    mask = ((torch.cumsum(is_eos.int(), 1) - is_eos.int()) == 0).float()
    return mdl.reinforce_update(video, s, mask, r, b, lr=1e-6, reuse_sampler_state=True)


for i in range(2):
    st = step(i)
torch.cuda.synchronize()
n = 3
t0 = time.time()
for i in range(n):
    st = step(i + 2)
torch.cuda.synchronize()
dt = (time.time() - t0) / n
print(f"B={B} K={K} Tc={Tc} V={V}: step {dt * 1e3:.1f} ms -> {K * B * Tc / dt:.0f} sampled tokens/s; loss {float(st.loss):.5f}; "
      f"grad norm {float(st.grad_sumsq.sqrt()):.4f}; peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GB")
from s2vt_amd import ops
ops.prof_filter(-1, -1); ops.prof_enable(True)
step(99); torch.cuda.synchronize()
rows = ops.prof_collect(); ops.prof_enable(False)
for r_ in sorted(rows, key=lambda r: -r["total_ms"]):
    print(f"  class {r_['kernel_class']} {r_['name']:<22} launches {r_['launches']:4d}  ms {r_['total_ms']:8.3f}  {r_['total_flops'] / r_['total_ms'] / 1e9:6.1f} TF")
