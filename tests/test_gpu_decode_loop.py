"""decode_loop.hip: the sampler's Tc decode steps -- LSTM2, vocabulary logits, multinomial / argmax pick -- in ONE persistent launch.
Since round 6 it is the product path at <= 64 decode rows (S2VT_DECLOOP unset / 1; its phase B runs on loader waves) and opt-in at 257-384 rows
(S2VT_DECLOOP=2, measured slower than the launches there).  Same chains, same keys: the token ids are those of the per-step launches
(S2VT_DECLOOP=0), bit for bit, at the bench dimensions (B = 64, K = 5: R = 384), at 320 rows (five row tiles per part), with a vocabulary that
leaves workgroups idle in the pick phase, with a batch that is not a power of two, and at 16 / 32 / 48 / 64 rows incl. a greedy-only batch."""
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
                           (32, 1, 12000, 1000, 500, 6), (16, 3, 9972, 1000, 500, 5), (16, 1, 2000, 992, 300, 4), (64, 0, 12000, 1000, 500, 5), (16, 0, 12000, 1000, 500, 20)):   # R <= 64: one row tile per row part
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


def test_persistent_decode_loop_draws_the_same_ids(gpu):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for flag in ("0", "2"):
            f = os.path.join(td, f"ids{flag}.npz")
            r = subprocess.run([sys.executable, "-c", CODE, f], env=dict(os.environ, S2VT_ROOT=root, S2VT_DECLOOP=flag), capture_output=True,
                               text=True, timeout=900)
            assert r.returncode == 0 and "child ok" in r.stdout, r.stderr[-3000:]
            res[flag] = dict(np.load(f))
    assert int(res["2"]["timeouts"]) == 0
    for k in res["0"]:
        assert np.array_equal(res["0"][k], res["2"][k]), k
    s = res["2"]["s64_5_12000_1000_11"]
    assert s.shape == (320, 7) and not np.array_equal(s, res["2"]["s64_5_12000_1000_12"])       # the noise stream matters
