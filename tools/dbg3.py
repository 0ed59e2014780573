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
p = {"lstm1_W": W, "lstm1_b": b}
vid = rng.integers(0, 1000, M).astype(np.int32); sid = rng.integers(0, 5, M).astype(np.int32)
keep, code = 1.0, 0
rc, rh, rout, rg, _ = oracle.lstm1_step(p, x, c, h, None, keep, want_gates=True)
for cfg in [int(a) for a in sys.argv[1:]]:
    bad = 0
    for it in range(30):
        gc, gh, gout, gg = gpu.lstm_cell_fwd(gpu.operand(_dev(x)), None, _dev(h), _dev(c), _dev(W), _dev(b), M, keep=keep,
                                             seed=77, video_id=_dev(vid), sample_id=_dev(sid), drop_code=code, want_gates=True, tile_cfg=cfg)
        g = gc.cpu().numpy()
        if not np.array_equal(g, rc):
            bad += 1
            d = np.argwhere(g != rc)
            if bad <= 2: print("  diff count", len(d), "rows", sorted(set(d[:, 0]))[:10], "cols", sorted(set(d[:, 1]))[:12], "maxabs", np.abs(g - rc).max())
    print("cfg", cfg, "bad", bad, "/30", flush=True)
