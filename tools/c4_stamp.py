"""dev: per-phase cycle split of lstm_chain4_kernel (build: tools/build_chain_variants.sh stamp:"-DS2VT_C4_STAMP";
run: S2VT_LIB=variants/lib_stamp.so python tools/c4_stamp.py)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import s2vt_amd
from s2vt_amd import ops
L = s2vt_amd.lib()
M, E, H, T = 320, 500, 1000, 25
rng = np.random.default_rng(0)
W = torch.as_tensor(rng.uniform(-.05, .05, (E + H, 4 * H)).astype(np.float32)).cuda(); b = torch.zeros(4 * H, device="cuda")
h0 = torch.zeros(M, H, device="cuda"); c0 = torch.zeros(M, H, device="cuda")
cinit = torch.as_tensor(rng.standard_normal((T, M, 4 * H)).astype(np.float32)).cuda()
vid = torch.arange(M, dtype=torch.int32, device="cuda"); sid = torch.zeros(M, dtype=torch.int32, device="cuda")
out = (C.c_ulonglong * 96)()
for rep in range(3):
    ops.lstm_recurrence_fwd(W, E, b, h0, c0, T=T, cinit=cinit, cinit_steps=T, keep=0.9, seed=1, video_id=vid, sample_id=sid, drop_code0=512,
                            want_gates=True, want_out=True, persistent=1)
    torch.cuda.synchronize()
    assert L.s2vt_c4_stamp_read(out) == 0
names = ["acc init", "grid wait", "first chunk", "chunk loop", "pointwise", "h stores+arrive", "history stores", "-"]
a = np.array(list(out), dtype=np.float64).reshape(3, 4, 8) / T
for w, wg in enumerate((0, 100, 251)):
    print("workgroup", wg)
    for i in range(7):
        print(f"  {names[i]:<18}", " ".join(f"{a[w, wave, i] / 100:8.1f}" for wave in range(4)), " (us at 100 MHz counter x?)")
    print("  total             ", " ".join(f"{a[w, wave].sum() / 100:8.1f}" for wave in range(4)))
