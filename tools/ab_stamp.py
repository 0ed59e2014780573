"""dev: per-phase clock split of attn_bwd_chain_kernel (build: tools/build_achain_variants.sh acstamp:"-DS2VT_AC_STAMP";
run: S2VT_LIB=variants/lib_acstamp.so python tools/ab_stamp.py)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import s2vt_amd
from s2vt_amd import attention as A, ops, hostglue
L = s2vt_amd.lib()
D, V, H, Tv, Tc, B = 1536, 12000, 1000, int(os.environ.get("TV", "5")), 20, 64
m = A.Attention_Caption_Generator(D, V, H, B, Tv, Tc, 0.9)
rng = np.random.default_rng(0)
video = torch.as_tensor(np.abs(rng.standard_normal((B, Tv, D)) * 0.5).astype(np.float32)).cuda()
cap = rng.integers(1, V, (B, Tc)).astype(np.int32)
mask = np.ones((B, Tc), np.float32)
out = (C.c_ulonglong * 48)()
for rep in range(3):
    m.xe_update(video, cap, mask, lr=0.0, active_steps=None)
    torch.cuda.synchronize()
    assert L.s2vt_ab_stamp_read(out) == 0
names = ["pointwise+dz images", "history", "dz wait", "MFMAs", "exchange", "dctx publish", "dctx wait+load", "dalpha", "de", "main+dhWa publish",
         "dhWa wait", "query product (rest)", "  q: fragments landed", "  q: MFMAs", "-", "-"]
a = np.array(list(out), dtype=np.float64).reshape(3, 16) / Tc / 100.0
print(f"{'phase (x100 clocks/step)':<26}" + "".join(f"{n:>12}" for n in ("first wg", "middle wg", "attn wg")))
for i in range(14):
    print(f"{names[i]:<26}" + "".join(f"{a[w, i]:12.2f}" for w in range(3)))
print(f"{'total':<26}" + "".join(f"{a[w].sum():12.2f}" for w in range(3)))
