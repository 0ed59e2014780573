"""dev: per-phase cycle split of decode_lstm4_kernel (tools/build_dec4_variants.sh stamp:"-DS2VT_D4_STAMP"; S2VT_LIB=variants/lib_stamp.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import s2vt_amd
from s2vt_amd import ops
L = s2vt_amd.lib()
from oracle import s2vt_oracle as orc
B, K = 64, 5
dims = ops.make_dims(1536, 12000, 500, 1000, 5, 20)
d = orc.Dims(1536, 12000, 500, 1000, 5, 20, 0)
p = {k: torch.as_tensor(v).cuda() for k, v in orc.init_params(d, 1).items()}
video = torch.rand(B, 5, 1536, device="cuda")
out = (C.c_ulonglong * 8)()
for rep in range(3):
    ops.sample(dims, ops.make_params(p), video, K, seed=1)
    torch.cuda.synchronize()
    L.s2vt_d4_stamp_read.argtypes = [C.c_void_p]
    assert L.s2vt_d4_stamp_read(out) == 0
a = np.array(list(out), dtype=np.float64) / 20
for n, v in zip(["prologue", "first chunk", "chunk loop", "pointwise", "stores"], a):
    print(f"{n:<12} {v:9.0f} cycles/launch")
print("total", a[:5].sum())
