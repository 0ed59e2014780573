import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import s2vt_amd
from s2vt_amd import ops, _lib
from oracle import s2vt_oracle as orc
B,K,V,H,E,Tc = 32,1,12000,1000,500,6
d = orc.Dims(256, V, E, H, 5, Tc, 0)
dims = ops.make_dims(256, V, E, H, 5, Tc)
p = {k: torch.as_tensor(v).cuda() for k, v in orc.init_params(d, 3).items()}
video = torch.as_tensor(np.abs(np.random.default_rng(B).standard_normal((B, 5, 256)) * 0.5).astype(np.float32)).cuda()
try:
    s, g = ops.sample(dims, ops.make_params(p), video, K, seed=11, video_base=7)
    torch.cuda.synchronize()
    print("ok", s.shape)
except Exception as e:
    print("ERR", e)
    L=_lib.lib()
    try:
        L.s2vt_last_hip_error.restype=__import__("ctypes").c_char_p
        print(L.s2vt_last_hip_error())
    except Exception as e2: print(e2)
