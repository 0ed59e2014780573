"""Dev helper: plain ([K][N]) vs transposed-W ([N][K]) form of the store contraction on the step's big shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import s2vt_amd
from s2vt_amd import ops

dev = "cuda"
torch.manual_seed(0)


def timeit(fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, K, N, nm) in [(6400, 1000, 12000, "logits"), (6400, 1500, 4000, "TF-decode"), (6400, 12000, 1000, "dO2"), (8000, 4000, 1500, "dX2"),
                      (384, 1000, 12000, "pick-shaped"), (320, 4000, 1000, "slab (no split)")]:
    A = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev); Wt = W.t().contiguous()
    for name, fn in (("NN", lambda c: ops.gemm([ops.operand(A)], W, None, M=M, tile_cfg=c)), ("NT", lambda c: ops.gemm_nt([ops.operand(A)], Wt, None, M=M, tile_cfg=c))):
        res = []
        for cfg in range(-1, 8):
            t = timeit(lambda: fn(cfg))
            res.append(f"cfg{cfg}:{t:.0f}us/{2 * M * K * N / t / 1e6:.0f}TF")
        print(f"{nm} {M}x{K}x{N} {name}: " + "  ".join(res), flush=True)
