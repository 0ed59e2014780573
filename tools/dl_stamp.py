"""dev: per-phase cycle split of decode_loop_kernel (tools/build_decloop_variants.sh stamp:"-DS2VT_DL_STAMP"; run with
S2VT_LIB=variants/lib_stamp.so S2VT_DECLOOP=1)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import s2vt_amd
from s2vt_amd import ops
L = s2vt_amd.lib()
from oracle import s2vt_oracle as orc
B, K, TC = int(os.environ.get("DL_B", "64")), int(os.environ.get("DL_K", "5")), 20
dims = ops.make_dims(1536, 12000, 500, 1000, 5, TC)
d = orc.Dims(1536, 12000, 500, 1000, 5, TC, 0)
p = {k: torch.as_tensor(v).cuda() for k, v in orc.init_params(d, 1).items()}
video = torch.rand(B, 5, 1536, device="cuda")
out = (C.c_ulonglong * 16)()
L.s2vt_dl_stamp_read.argtypes = [C.c_void_p]
for rep in range(3):
    ops.sample(dims, ops.make_params(p), video, K, seed=1)
    torch.cuda.synchronize()
    assert L.s2vt_dl_stamp_read(out) == 0
a = np.array(list(out), dtype=np.float64)
names = ["prologue (once)", "A: tokens+partial+chunk loop", "A: pointwise + image store", "A->B arrive + wait", "B: main loop", "B: pick epilogue", "B->A arrive + wait"]
for n, v in zip(names, a):
    print(f"{n:<32} {v / TC:9.0f} shader cycles/step  ({v / TC / 2320:.1f} us at 2.32 GHz)")
for i, n in ((8, "loader: rest of the step"), (9, "loader: stage issue"), (10, "loader: data wait"), (11, "loader: barrier")):
    print(f"{n:<32} {a[i] / TC:9.0f} shader cycles/step  ({a[i] / TC / 2320:.1f} us)")
print("total per step", a[1:7].sum() / TC / 2320, "us; timeouts", ops.chain_timeouts())
