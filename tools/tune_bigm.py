"""Dev helper: tile configurations of the step kernels at the reference's default batch (M = 2304 sampler rows, 2048 loss rows)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import s2vt_amd
from s2vt_amd import ops

dev = "cuda"; H, E, V = 1000, 500, 9972
torch.manual_seed(0)


def timeit(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


W2 = torch.randn(2 * H + E, 4 * H, device=dev) * 0.03; b2 = torch.zeros(4 * H, device=dev)
Wemb = torch.randn(V, E, device=dev) * 0.1
Wout = torch.randn(H, V, device=dev) * 0.1; bout = torch.zeros(V, device=dev)
for M in (int(x) for x in os.environ.get("MS", "2304,2048,256").split(",")):
    h = torch.randn(M, H, device=dev); c = torch.randn(M, H, device=dev)
    idx = torch.randint(0, V, (M,), device=dev, dtype=torch.int32)
    for name, x0, x1 in (("K=1500", ops.operand(None, k=H), ops.operand(Wemb, rowidx=idx)), ("K=1000", ops.operand(None, k=H + E), None)):
        flops = 2 * M * (1500 if x1 is not None else 1000) * 4 * H
        res = []
        for cfg in range(-1, int(os.environ.get('NCFG', '12'))):
            t = timeit(lambda: ops.lstm_cell_fwd(x0, x1, h, c, W2, b2, M, tile_cfg=cfg))
            res.append(f"cfg{cfg}:{t:.0f}us/{flops / t / 1e6:.0f}TF")
        print(f"LSTM2 M={M} {name}: " + "  ".join(res), flush=True)
M = 2304
vid = torch.zeros(M, dtype=torch.int32, device=dev); sid = torch.zeros(M, dtype=torch.int32, device=dev)
o2 = torch.randn(M, H, device=dev)
res = []
for cfg in range(-1, 6):
    t = timeit(lambda: ops.vocab_pick(o2, Wout, bout, vid, sid, 0, 1, tile_cfg=cfg))
    res.append(f"cfg{cfg}:{t:.0f}us/{2 * M * H * V / t / 1e6:.0f}TF")
print(f"PICK M={M}: " + "  ".join(res), flush=True)
