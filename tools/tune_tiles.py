"""Dev helper: time every tile configuration of the forward kernels on the bench shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import s2vt_amd
from s2vt_amd import ops

dev = "cuda"
H, E, V = 1000, 500, 12000
torch.manual_seed(0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


W2 = torch.randn(2 * H + E, 4 * H, device=dev) * 0.03; b2 = torch.zeros(4 * H, device=dev)
W1 = torch.randn(E + H, 4 * H, device=dev) * 0.03
Wemb = torch.randn(V, E, device=dev) * 0.1
Wout = torch.randn(H, V, device=dev) * 0.1; bout = torch.zeros(V, device=dev)
for M in (384, 320, 64):
    h = torch.randn(M, H, device=dev); c = torch.randn(M, H, device=dev); o1 = torch.randn(M, H, device=dev)
    idx = torch.randint(0, V, (M,), device=dev, dtype=torch.int32)
    for name, x0, x1, W in (("LSTM2 K=2500", ops.operand(o1), ops.operand(Wemb, rowidx=idx), W2), ("LSTM1 K=1000", ops.operand(None, k=E), None, W1)):
        flops = 2 * M * (W.shape[0] - (E if x1 is None else 0)) * 4 * H
        res = []
        for cfg in range(12):
            try:
                t = timeit(lambda: ops.lstm_cell_fwd(x0, x1, h, c, W, b2, M, tile_cfg=cfg))
            except Exception:
                break
            res.append(f"cfg{cfg}:{t:.0f}us/{flops / t / 1e6:.0f}TF")
        print(f"M={M} {name}: " + "  ".join(res), flush=True)
vid = torch.zeros(384, dtype=torch.int32, device=dev); sid = torch.zeros(384, dtype=torch.int32, device=dev)
o2 = torch.randn(384, H, device=dev)
res = []
for cfg in range(12):
    try:
        t = timeit(lambda: ops.vocab_pick(o2, Wout, bout, vid, sid, 0, 1, tile_cfg=cfg))
    except Exception:
        break
    res.append(f"cfg{cfg}:{t:.0f}us/{2 * 384 * H * V / t / 1e6:.0f}TF")
print("PICK M=384: " + "  ".join(res), flush=True)
sidg = -torch.ones(384, dtype=torch.int32, device=dev)
t = timeit(lambda: ops.vocab_pick(o2, Wout, bout, vid, sidg, 0, 1)); print(f"PICK greedy-only rows (no gumbel): {t:.0f}us")
for (M, K, N, nm) in ((6400, 1000, 12000, "logits"), (6400, 12000, 1000, "dO2"), (8000, 4000, 1500, "dX2"), (320, 4000, 1000, "dh slab (no split)")):
    A = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev)
    res = []
    for cfg in range(12):
        try:
            t = timeit(lambda: ops.gemm([ops.operand(A)], W, None, M=M, tile_cfg=cfg), n=5)
        except Exception:
            break
        res.append(f"cfg{cfg}:{t:.0f}us/{2 * M * K * N / t / 1e6:.0f}TF")
    print(f"STORE {nm} {M}x{K}x{N}: " + "  ".join(res), flush=True)
