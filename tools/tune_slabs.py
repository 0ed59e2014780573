"""Dev helper: tile x split-K sweep of the data-gradient products (and the forward products beside them) at the row counts
a truncated XE / small-batch unroll produces (64 rows per step).  Prints the best (cfg, splits) per shape and what the
library's own choice costs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import s2vt_amd
from s2vt_amd import ops

dev = "cuda"
torch.manual_seed(0)


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


rows = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [384, 640, 896, 1216, 1600]
for (nm, N, K) in [("dO2", 1000, 12000), ("dX2", 1500, 4000)]:
    Wt = torch.randn(N, K, device=dev)
    for M in rows:
        A = torch.randn(M, K, device=dev)
        slabs = torch.empty(13 * M * N, device=dev)
        t_auto = timeit(lambda: ops.gemm_nt_splitk(A, Wt, 0, -1, slabs=slabs))
        res = []
        for cfg in (0, 1, 2, 5, 6, 7):
            for s in (1, 2, 3, 4, 6, 8, 12):
                if K // s < 256:
                    continue
                res.append((timeit(lambda: ops.gemm_nt_splitk(A, Wt, s, cfg, slabs=slabs)), cfg, s))
        res.sort()
        print(f"{nm} M={M} N={N} K={K}: library {t_auto:.0f} us ({2 * M * N * K / t_auto / 1e6:.0f} TF) | best " +
              "  ".join(f"cfg{c}/s{s}:{t:.0f}" for t, c, s in res[:5]), flush=True)
for (nm, N, K) in [("logits", 12000, 1000), ("hoisted", 4000, 1500)]:
    W = torch.randn(K, N, device=dev)
    for M in rows:
        A = torch.randn(M, K, device=dev)
        t_auto = timeit(lambda: ops.gemm([ops.operand(A)], W, None, M=M))
        res = sorted((timeit(lambda: ops.gemm([ops.operand(A)], W, None, M=M, tile_cfg=c)), c) for c in range(8))
        print(f"{nm} M={M} N={N} K={K}: library {t_auto:.0f} us ({2 * M * N * K / t_auto / 1e6:.0f} TF) | best " +
              "  ".join(f"cfg{c}:{t:.0f}" for t, c in res[:4]), flush=True)
