"""PICK (vocab logits + Gumbel-max) tile sweep across M: GPU time per launch from the in-library HIP-event profiler
(wall time around the Python call is CPU-bound below ~40 us).  env: H, V, MS (comma list)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, s2vt_amd
from s2vt_amd import ops
dev = "cuda"; H, V = int(os.environ.get('H', '1000')), int(os.environ.get('V', '12000'))
torch.manual_seed(0)
Wout = torch.randn(H, V, device=dev) * 0.1; bout = torch.zeros(V, device=dev)
ncfg = int(os.environ.get('NCFG', '7'))
for M in [int(x) for x in os.environ.get('MS', '32,64,96,128,192,256,320,384').split(',')]:
    vid = torch.zeros(M, dtype=torch.int32, device=dev); sid = torch.zeros(M, dtype=torch.int32, device=dev)
    o2 = torch.randn(M, H, device=dev)
    res = []
    for cfg in [int(x) for x in os.environ['CFGS'].split(',')] if 'CFGS' in os.environ else range(ncfg):
        # clocks: an idle GPU between Python calls runs these launches ~50 % slower than the step does -- keep the queue full
        for _ in range(300): ops.vocab_pick(o2, Wout, bout, vid, sid, 0, 1, tile_cfg=cfg)
        torch.cuda.synchronize()
        ops.prof_filter(-1, -1); ops.prof_enable(True)
        for _ in range(100): ops.vocab_pick(o2, Wout, bout, vid, sid, 0, 1, tile_cfg=cfg)
        torch.cuda.synchronize()
        ops.prof_enable(False)
        rows = [r for r in ops.prof_collect() if r["kernel_class"] == 2]
        t = sum(r["total_ms"] for r in rows) / max(1, sum(r["launches"] for r in rows)) * 1e3
        res.append(f"{rows[0]['name'] if rows else cfg}:{t:.0f}us/{2*M*H*V/t/1e6:.0f}TF")
    print(f"PICK M={M}: " + "  ".join(res), flush=True)
