import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import s2vt_amd
from s2vt_amd import ops
M, H, E = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(0)
W = torch.randn(E + H, 4 * H, device="cuda") * 0.1; b = torch.zeros(4 * H, device="cuda")
x = torch.randn(M, E, device="cuda"); h = torch.randn(M, H, device="cuda"); c = torch.randn(M, H, device="cuda")
vid = torch.arange(M, dtype=torch.int32, device="cuda"); sid = torch.zeros(M, dtype=torch.int32, device="cuda")
ref = ops.lstm_cell_fwd(ops.operand(x), None, h, c, W, b, M, tile_cfg=0, want_gates=True)
torch.cuda.synchronize()
ops.prof_enable(True)
ops.lstm_cell_fwd(ops.operand(x), None, h, c, W, b, M, tile_cfg=-1)
torch.cuda.synchronize(); ops.prof_enable(False)
print("auto config:", [r["name"] for r in ops.prof_collect()])
for cfg in [int(a) for a in sys.argv[4:]]:
    bad = 0
    for it in range(12):
        time.sleep(0.25)
        out = ops.lstm_cell_fwd(ops.operand(x), None, h, c, W, b, M, tile_cfg=cfg, want_gates=True)
        torch.cuda.synchronize()
        ok = all(torch.equal(a, r) for a, r in zip(out, ref))
        if not ok:
            bad += 1
            d = (out[0] != ref[0]).nonzero()
            if bad == 1: print("  first diff idx", d[:4].tolist(), "count", d.shape[0])
    print("cfg", cfg, "bad runs", bad, "/ 12", flush=True)
