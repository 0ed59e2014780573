"""Dev helper: launch one kernel class repeatedly (for rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import s2vt_amd
from s2vt_amd import ops
dev = "cuda"; H, E, V = 1000, 500, 12000
which = sys.argv[1] if len(sys.argv) > 1 else "lstm"
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else -1
M = int(sys.argv[3]) if len(sys.argv) > 3 else 384
torch.manual_seed(0)
if which == "lstm":
    W2 = torch.randn(2 * H + E, 4 * H, device=dev) * 0.03; b2 = torch.zeros(4 * H, device=dev)
    Wemb = torch.randn(V, E, device=dev) * 0.1
    h = torch.randn(M, H, device=dev); c = torch.randn(M, H, device=dev); o1 = torch.randn(M, H, device=dev)
    idx = torch.randint(0, V, (M,), device=dev, dtype=torch.int32)
    x0, x1 = ops.operand(o1), ops.operand(Wemb, rowidx=idx)
    for _ in range(12):
        ops.lstm_cell_fwd(x0, x1, h, c, W2, b2, M, tile_cfg=cfg)
elif which == "pick":
    Wout = torch.randn(H, V, device=dev) * 0.1; bout = torch.zeros(V, device=dev)
    vid = torch.zeros(M, dtype=torch.int32, device=dev); sid = torch.zeros(M, dtype=torch.int32, device=dev)
    o2 = torch.randn(M, H, device=dev)
    for _ in range(12):
        ops.vocab_pick(o2, Wout, bout, vid, sid, 0, 1, tile_cfg=cfg)
else:
    A = torch.randn(6400, 1000, device=dev); W = torch.randn(1000, 12000, device=dev)
    for _ in range(6):
        ops.gemm([ops.operand(A)], W, None, M=6400, tile_cfg=cfg)
torch.cuda.synchronize()
