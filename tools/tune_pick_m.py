import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, s2vt_amd
from s2vt_amd import ops
dev="cuda"; H,V=1000,9972
torch.manual_seed(0)
Wout = torch.randn(H, V, device=dev) * 0.1; bout = torch.zeros(V, device=dev)
def timeit(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (512, 768, 1024, 1536, 2048, 2304):
    vid = torch.zeros(M, dtype=torch.int32, device=dev); sid = torch.zeros(M, dtype=torch.int32, device=dev)
    o2 = torch.randn(M, H, device=dev)
    res=[]
    for cfg in range(0, 6):
        t = timeit(lambda: ops.vocab_pick(o2, Wout, bout, vid, sid, 0, 1, tile_cfg=cfg))
        res.append(f"cfg{cfg}:{t:.0f}us/{2*M*H*V/t/1e6:.0f}TF")
    print(f"PICK M={M}: " + "  ".join(res), flush=True)
