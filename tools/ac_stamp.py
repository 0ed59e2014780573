"""dev: per-phase clock split of attn_chain_kernel (build: tools/build_achain_variants.sh acstamp:"-DS2VT_AC_STAMP";
run: S2VT_LIB=variants/lib_acstamp.so python tools/ac_stamp.py)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import s2vt_amd
from s2vt_amd import attention as A, ops
L = s2vt_amd.lib()
D, V, H, Tv, Tc, B = 1536, 12000, 1000, int(os.environ.get("TV", "5")), 20, 64
m = A.Attention_Caption_Generator(D, V, H, B, Tv, Tc, 0.9)
rng = np.random.default_rng(0)
video = torch.as_tensor(np.abs(rng.standard_normal((B, Tv, D)) * 0.5).astype(np.float32)).cuda()
cap = torch.as_tensor(rng.integers(0, V, (B, Tc)).astype(np.int32)).cuda()
vid, sid = m._row_ids(B)
out = (C.c_ulonglong * 48)()
fn = getattr(L, os.environ.get("STAMP_FN", "s2vt_ac_stamp_read"))
for rep in range(3):
    ops.attn_teacher_forced_fwd(m.dims, m.store.params, video, cap, 0.9, 77, vid, sid)
    torch.cuda.synchronize()
    assert fn(out) == 0
names = ["partial", "step wait", "query+publish", "h block", "hWa wait+load", "tanh+chains", "softmax+ctx+publish", "ctx wait", "ctx block",
         "pointwise+arrive", "history", "-", "-", "-", "-", "-"]
a = np.array(list(out), dtype=np.float64).reshape(3, 16) / Tc / 100.0          # us per step (100 MHz counter)
print(f"{'phase':<22}" + "".join(f"{n:>12}" for n in ("query wg", "plain wg", "attn wg")))
for i in range(11):
    print(f"{names[i]:<22}" + "".join(f"{a[w, i]:12.2f}" for w in range(3)))
print(f"{'total':<22}" + "".join(f"{a[w].sum():12.2f}" for w in range(3)))
