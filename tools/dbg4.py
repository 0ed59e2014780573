import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import s2vt_amd
from s2vt_amd import ops as gpu
from oracle import s2vt_oracle as oracle
def _dev(a): return torch.as_tensor(np.ascontiguousarray(a)).cuda()
M, E, H = 96, 500, 1000
rng = np.random.default_rng(M + H)
W = rng.uniform(-.3, .3, (E + H, 4 * H)).astype(np.float32); b = rng.uniform(-.5, .5, 4 * H).astype(np.float32)
x = rng.standard_normal((M, E)).astype(np.float32); c = rng.standard_normal((M, H)).astype(np.float32)
h = rng.uniform(-1, 1, (M, H)).astype(np.float32)
vid = rng.integers(0, 1000, M).astype(np.int32); sid = rng.integers(0, 5, M).astype(np.int32)
rc, rh, rout, rg, _ = oracle.lstm1_step({"lstm1_W": W, "lstm1_b": b}, x, c, h, None, 1.0, want_gates=True)
dx, dh, dc, dW, db, dvid, dsid = _dev(x), _dev(h), _dev(c), _dev(W), _dev(b), _dev(vid), _dev(sid)
mode = sys.argv[1]
for cfg in (0, 5):
    if mode == "persist":
        out = gpu.lstm_cell_fwd(gpu.operand(dx), None, dh, dc, dW, db, M, tile_cfg=cfg, want_gates=True)
    elif mode == "persist_ids":
        out = gpu.lstm_cell_fwd(gpu.operand(dx), None, dh, dc, dW, db, M, keep=1.0, seed=77, video_id=dvid, sample_id=dsid, drop_code=0, want_gates=True, tile_cfg=cfg)
    elif mode == "temp":
        out = gpu.lstm_cell_fwd(gpu.operand(_dev(x)), None, _dev(h), _dev(c), _dev(W), _dev(b), M, tile_cfg=cfg, want_gates=True)
    torch.cuda.synchronize()
    g = out[0].cpu().numpy()
    print(mode, "cfg", cfg, "equal oracle:", np.array_equal(g, rc), "ndiff", int((g != rc).sum()), flush=True)
    if not np.array_equal(g, rc):
        d = np.argwhere(g != rc)
        print("   rows", sorted(set(d[:, 0].tolist())), "cols", sorted(set(d[:, 1].tolist()))[:40])
        gg = out[3].cpu().numpy(); dg = np.argwhere(gg != rg); print("   gates diff", len(dg), sorted(set((dg[:,1]//H).tolist())), sorted(set((dg[:,1]%H).tolist()))[:20])
