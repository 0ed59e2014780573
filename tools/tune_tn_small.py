"""Dev helper: the weight-gradient (TN) contractions of the XE / multitask steps (short reductions, 896-1600 rows) one by one: us and TFLOP/s per shape,
under the slab-count knob S2VT_TN_WGS (workgroups wanted; 0 = the cost rule)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import s2vt_amd
from s2vt_amd import ops


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


dev = "cuda"
torch.manual_seed(0)
for (Mred, Kout, N, nm) in [(896, 1000, 12000, "dWout XE (14 steps x 64)"), (1216, 1000, 4000, "dW2.h2 / dW2.out1 / dW1.h XE (19 x 64)"),
                            (1280, 1000, 12000, "dWout multitask (20 x 64)"), (1600, 1000, 4000, "dW2.* multitask (25 x 64)"), (896, 500, 4000, "dW2.emb XE"),
                            (6400, 1000, 12000, "dWout rl (20 x 320)"), (8000, 1000, 4000, "dW2.* rl (25 x 320)")]:
    A = torch.randn(Mred, Kout, device=dev); Bm = torch.randn(Mred, N, device=dev); C = torch.zeros(Kout, N, device=dev)
    us = timeit(lambda: ops.gemm_tn(A, Bm, C, accumulate=False))
    print(f"{nm:<44} Mred={Mred:5d} {Kout}x{N}: {us:7.1f} us  {2.0 * Mred * Kout * N / us / 1e6:6.1f} TFLOP/s", flush=True)
