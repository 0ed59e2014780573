"""Dev helper: the M <= 64 recurrent LSTM step (LSTM1 form: hoisted input partial + h @ W[E:], K = 1000) back to back,
as the encode / decode loops launch it -- per-launch time of every gw tile, for A/B builds (S2VT_LIB)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import s2vt_amd
from s2vt_amd import ops

dev = "cuda"
H, E = 1000, 500
torch.manual_seed(0)
W1 = torch.randn(E + H, 4 * H, device=dev) * 0.03; b = torch.zeros(4 * H, device=dev)
for M in (64, 32):
    hs = [torch.randn(M, H, device=dev) for _ in range(2)]; cs = [torch.randn(M, H, device=dev) for _ in range(2)]
    res = []
    for cfg in (-1, 3, 4, 6, 9):
        def chain(n=25):
            for t in range(n):
                ops.lstm_cell_fwd(ops.operand(None, k=E), None, hs[t & 1], cs[t & 1], W1, b, M, tile_cfg=cfg)
        chain(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); chain(100); e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 100 * 1e3
        res.append(f"cfg{cfg}:{t:.1f}us/{2 * M * H * 4 * H / t / 1e6:.0f}TF")
    print(f"M={M} LSTM1 step K=1000: " + "  ".join(res), flush=True)
