"""dev: one lstm_chain4_kernel launch set (M = 320, H = 1000, T = 25) for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import s2vt_amd
from s2vt_amd import ops
M, E, H, T = 320, 500, 1000, 25
rng = np.random.default_rng(0)
W = torch.as_tensor(rng.uniform(-.05, .05, (E + H, 4 * H)).astype(np.float32)).cuda(); b = torch.zeros(4 * H, device="cuda")
h0 = torch.zeros(M, H, device="cuda"); c0 = torch.zeros(M, H, device="cuda")
cinit = torch.as_tensor(rng.standard_normal((T, M, 4 * H)).astype(np.float32)).cuda()
vid = torch.arange(M, dtype=torch.int32, device="cuda"); sid = torch.zeros(M, dtype=torch.int32, device="cuda")
for rep in range(3):
    ops.lstm_recurrence_fwd(W, E, b, h0, c0, T=T, cinit=cinit, cinit_steps=T, keep=0.9, seed=1, video_id=vid, sample_id=sid, drop_code0=512,
                            want_gates=True, want_out=True, persistent=1)
torch.cuda.synchronize()
