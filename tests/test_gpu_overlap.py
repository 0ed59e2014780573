"""Gated overlap (csrc/train.hip and attn_model.hip, S2VT_OVERLAP=2, the default): at <= 256 unrolled rows the weight-gradient contractions that do not feed a
recurrence -- dWout beside LSTM2's backward recurrence, LSTM2's three beside LSTM1's -- run on a side stream, released by a gate once the
persistent grid is resident.  Same kernels, same operands, another ORDER of launches on two queues: the gradients must equal the single-stream
one (S2VT_OVERLAP=0) up to the order-free reductions' noise (split-K atomics), with no grid-wide wait timing out, also when the step is
repeated back to back (the side stream of step i must have joined before step i + 1 touches the bucket)."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CODE = r'''
import os, sys
sys.path.insert(0, os.environ["S2VT_ROOT"])
import numpy as np, torch
import s2vt_amd
from s2vt_amd import model as M, hostglue, ops
out = {}
for (B, K, V, H, E, Tc) in ((64, 0, 3000, 1000, 500, 8), (32, 1, 2000, 1000, 500, 6), (16, 3, 1000, 256, 64, 5)):
    mdl = M.Video_Caption_Generator(256, V, E, H, B, 0, 5, Tc, seed=5, multisample=max(K, 1))
    rng = np.random.default_rng(B)
    video = torch.as_tensor(np.abs(rng.standard_normal((B, 5, 256)) * 0.5).astype(np.float32)).cuda()
    cap = rng.integers(2, V, (B, Tc)).astype(np.int32); cap[:, -1] = 0
    mask = hostglue.masks_from_ids(cap)
    for rep in range(3):                                    # back to back, lr = 0 (an Adam step would turn noise-level gradient differences into sign flips):
                                                            # a step that zeroed the bucket under the previous step's side stream would show in the gradients
        if K == 0:
            st = mdl.xe_update(video, cap, mask, lr=0.0, q1=True)
        else:
            s, _ = mdl.sample(video, K, True, seed=3)
            sm = ops.caption_mask(s, want_target=False)[0]
            r = torch.as_tensor(rng.random(K * B).astype(np.float32)).cuda(); b = torch.as_tensor(np.tile(rng.random(B).astype(np.float32), K)).cuda()
            st = mdl.mixed_update(video, s, sm, r, b, cap, mask, lr=0.0, lambda_loss=0.5) if K == 1 else mdl.reinforce_update(video, s, sm, r, b, lr=0.0)
    torch.cuda.synchronize()
    out[f"theta_{B}_{K}"] = mdl.store.grad[:mdl.store.numel].cpu().numpy(); out[f"loss_{B}_{K}"] = np.asarray(float(st.loss)); out[f"gn_{B}_{K}"] = np.asarray(float(st.grad_sumsq))
# the temporal-attention captioner (attn_model.hip): dWout and the output layer's three weight gradients behind the persistent backward recurrence (Tv <= 5)
from s2vt_amd import attention as A
for (B, Tv, V, H, Tc) in ((64, 5, 3000, 1000, 8), (16, 3, 500, 64, 5)):
    am = A.Attention_Caption_Generator(128, V, H, B, Tv, Tc, 0.9, seed=5)
    rng = np.random.default_rng(100 + B)
    video = torch.as_tensor(np.abs(rng.standard_normal((B, Tv, 128)) * 0.5).astype(np.float32)).cuda()
    cap = rng.integers(2, V, (B, Tc)).astype(np.int32); cap[:, -1] = 0
    mask = hostglue.masks_from_ids(cap)
    for rep in range(3):
        st = am.xe_update(video, cap, mask, lr=0.0, keep=0.9)
    torch.cuda.synchronize()
    out[f"theta_att_{B}"] = am.store.grad[:am.store.numel].cpu().numpy(); out[f"loss_att_{B}"] = np.asarray(float(st.loss)); out[f"gn_att_{B}"] = np.asarray(float(st.grad_sumsq))
out["timeouts"] = np.asarray(ops.chain_timeouts())
np.savez(sys.argv[1], **out)
print("child ok")
'''


def test_gated_overlap_equals_the_single_stream_update(gpu):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for mode in ("0", "2"):
            f = os.path.join(td, f"o{mode}.npz")
            env = dict(os.environ, S2VT_ROOT=root, S2VT_OVERLAP=mode)
            r = subprocess.run([sys.executable, "-c", CODE, f], env=env, capture_output=True, text=True, timeout=900)
            assert r.returncode == 0 and "child ok" in r.stdout, r.stderr[-3000:]
            res[mode] = dict(np.load(f))
    assert int(res["0"]["timeouts"]) == 0 and int(res["2"]["timeouts"]) == 0
    for k in res["0"]:
        if k.startswith("theta"):
            a, b = res["0"][k], res["2"][k]
            assert np.isfinite(b).all()
            assert np.abs(a).max() > 0 and np.abs(a - b).max() <= 2e-5 * np.abs(a).max(), (k, float(np.abs(a - b).max()), float(np.abs(a).max()))
        elif k.startswith(("loss", "gn")):
            assert abs(float(res["0"][k]) - float(res["2"][k])) <= 2e-3 * max(1.0, abs(float(res["0"][k]))), k
