"""Dev helper: every STORE tile configuration on the batched (hoisted) contractions of the bench step."""
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
    return e0.elapsed_time(e1) / n * 1e3   # us


shapes = [(1600, 1000, 4000, "P2 / TF-encode"), (6400, 1500, 4000, "TF-decode"), (8000, 4000, 1500, "dX2"), (6400, 1000, 12000, "logits"),
          (6400, 12000, 1000, "dO2"), (320, 500, 4000, "Xp1"), (320, 1536, 500, "frame embed"), (320, 4000, 500, "dX1")]
for (M, K, N, nm) in shapes:
    A = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev)
    res = []
    for cfg in (-1, 2, 6, 8, 9, 10, 11):
        try:
            t = timeit(lambda: ops.gemm([ops.operand(A)], W, None, M=M, tile_cfg=cfg))
        except Exception as e:
            res.append(f"cfg{cfg}:ERR")
            continue
        res.append(f"cfg{cfg}:{t:.0f}us/{2 * M * K * N / t / 1e6:.0f}TF")
    print(f"STORE {nm} {M}x{K}x{N}: " + "  ".join(res), flush=True)
