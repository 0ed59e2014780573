"""Dev helper: us per launch of the sampler step's two kernels, register-staged tiles against the LDS-DMA ring tiles (tile_cfg indices of fwd.hip)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import s2vt_amd
from s2vt_amd import ops


def timeit(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


dev = "cuda"
H, E, V = 1000, 500, 12000
torch.manual_seed(0)
W2 = torch.randn(2 * H + E, 4 * H, device=dev) * 0.03; b2 = torch.zeros(4 * H, device=dev)
Wemb = torch.randn(V, E, device=dev) * 0.1
Wout = torch.randn(H, V, device=dev) * 0.1; bout = torch.zeros(V, device=dev)
LSTM = {384: (11, 12, 17, 18), 64: (9, 13, 14), 128: (4, 15), 256: (6, 16)}
PICK = {384: (4, 7, 8, 11, 12, 14), 64: (6, 9, 10, 13), 128: (6, 9, 13), 256: (6, 9, 4, 11)}
for M, cfgs in LSTM.items():
    h = torch.randn(M, H, device=dev); c = torch.randn(M, H, device=dev)
    idx = torch.randint(0, V, (M,), device=dev, dtype=torch.int32)
    for cfg in cfgs:
        us = timeit(lambda: ops.lstm_cell_fwd(ops.operand(None, k=H), ops.operand(Wemb, rowidx=idx), h, c, W2, b2, M, tile_cfg=cfg))
        print(f"cell M={M} cfg {cfg}: {us:.1f} us  {2.0 * M * 1500 * 4000 / us / 1e6:.1f} TFLOP/s", flush=True)
for M, cfgs in PICK.items():
    vid = torch.zeros(M, dtype=torch.int32, device=dev); sid = torch.zeros(M, dtype=torch.int32, device=dev)
    o2 = torch.randn(M, H, device=dev)
    for pc in cfgs:
        us = timeit(lambda: ops.vocab_pick(o2, Wout, bout, vid, sid, 0, 1, tile_cfg=pc))
        print(f"pick M={M} cfg {pc}: {us:.1f} us  {2.0 * M * H * V / us / 1e6:.1f} TFLOP/s", flush=True)
