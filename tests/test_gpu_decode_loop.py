"""(Experimental build only: `make -C multitask-end-to-end-video-captioning_amd/csrc EXPERIMENTAL=1` -> libs2vt_hip_experimental.so; skipped when that
library has not been built -- the kernel is a recorded negative result, DESIGN.md 11, and not part of the product library.)
decode_loop.hip (S2VT_DECLOOP=1): the sampler's Tc decode steps -- LSTM2, vocabulary logits, multinomial / argmax pick --
in ONE persistent launch.  Same chains, same keys: the token ids are those of the per-step launches, bit for bit, at the
bench dimensions (B = 64, K = 5: R = 384), at 320 rows (five row tiles per part), with a vocabulary that leaves workgroups
idle in the pick phase, and with a batch that is not a power of two."""
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
from s2vt_amd import ops
from oracle import s2vt_oracle as orc
out = {}
for (B, K, V, H, E, Tc) in ((64, 5, 12000, 1000, 500, 7), (64, 4, 12000, 1000, 500, 4), (48, 6, 9972, 1000, 500, 5), (64, 5, 2000, 992, 300, 5),
                           (32, 1, 12000, 1000, 500, 6), (16, 3, 9972, 1000, 500, 5), (16, 1, 2000, 992, 300, 4), (64, 0, 12000, 1000, 500, 5)):   # R <= 64: one row tile per row part
    d = orc.Dims(256, V, E, H, 5, Tc, 0)
    dims = ops.make_dims(256, V, E, H, 5, Tc)
    p = {k: torch.as_tensor(v).cuda() for k, v in orc.init_params(d, 3).items()}
    p["embed_word_b"] = torch.as_tensor(np.random.default_rng(1).uniform(-.5, .5, V).astype(np.float32)).cuda()
    p["lstm2_b"] = torch.as_tensor(np.random.default_rng(2).uniform(-.1, .1, 4 * H).astype(np.float32)).cuda()
    video = torch.as_tensor(np.abs(np.random.default_rng(B).standard_normal((B, 5, 256)) * 0.5).astype(np.float32)).cuda()
    for seed in (11, 12):
        s, g = ops.sample(dims, ops.make_params(p), video, K, seed=seed, video_base=7)
        out[f"s{B}_{K}_{V}_{H}_{seed}"] = s.cpu().numpy(); out[f"g{B}_{K}_{V}_{H}_{seed}"] = g.cpu().numpy()
out["timeouts"] = np.asarray(ops.chain_timeouts())
np.savez(sys.argv[1], **out)
print("child ok")
'''


def experimental_lib():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = os.path.join(root, "multitask-end-to-end-video-captioning_amd", "libs2vt_hip_experimental.so")
    if not os.path.exists(p):
        pytest.skip("libs2vt_hip_experimental.so not built (make EXPERIMENTAL=1): the opt-in decode-loop experiments are not in the product library")
    return p


def test_persistent_decode_loop_draws_the_same_ids(gpu):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = experimental_lib()
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for flag in ("0", "1"):
            f = os.path.join(td, f"ids{flag}.npz")
            r = subprocess.run([sys.executable, "-c", CODE, f], env=dict(os.environ, S2VT_ROOT=root, S2VT_DECLOOP=flag, S2VT_LIB=lib), capture_output=True,
                               text=True, timeout=900)
            assert r.returncode == 0 and "child ok" in r.stdout, r.stderr[-3000:]
            res[flag] = dict(np.load(f))
    assert int(res["1"]["timeouts"]) == 0
    for k in res["0"]:
        assert np.array_equal(res["0"][k], res["1"][k]), k
    s = res["1"]["s64_5_12000_1000_11"]
    assert s.shape == (320, 7) and not np.array_equal(s, res["1"]["s64_5_12000_1000_12"])       # the noise stream matters
