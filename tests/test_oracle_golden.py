"""The frozen oracle outputs (tests/golden/oracle_small.npz, tools/make_oracle_golden.py): the CPU
oracle must keep reproducing them bit-for-bit, and (GPU) so must the HIP path."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_small.npz"))


def test_oracle_reproduces_golden(oracle):
    import make_oracle_golden as mk
    cur = mk.build()
    assert sorted(cur) == sorted(G.files)
    for k in G.files:
        assert np.array_equal(np.asarray(cur[k]), G[k]), k


@pytest.mark.gpu
def test_hip_reproduces_golden(gpu):
    import torch
    import make_oracle_golden as mk
    from s2vt_amd import model as M
    d = mk.DIMS
    mdl = M.Video_Caption_Generator(d["dim_image"], d["n_words"], d["word_dim"], d["lstm_dim"], 3, 0, d["n_video_lstm_step"],
                                    d["n_caption_lstm_step"])
    mdl.store.load({k[6:]: G[k] for k in G.files if k.startswith("param_")})
    s, g = mdl.sample(G["video"], 2, True, seed=77, video_base=2)
    assert np.array_equal(s.cpu().numpy(), G["sampled"]) and np.array_equal(g.cpu().numpy(), G["greedy"])
    dev = lambda a, t: torch.as_tensor(a).to("cuda", t)
    vid, sid = mdl._row_ids(3, 2, 2)
    logits, _ = gpu.teacher_forced_fwd(mdl.dims, mdl.store.params, dev(G["video"], torch.float32), dev(G["sampled"], torch.int32), 6, 0.9,
                                       501, vid, sid)
    got = logits.view(5, 6, -1).permute(1, 0, 2).cpu().numpy()
    assert np.array_equal(got, G["tf_logits"])
    coef = dev((G["mask"] * (G["rewards"] - G["baseline"])[:, None]).T.copy(), torch.float32).reshape(-1)
    nll, _ = gpu.softmax_nll_fwd_bwd(logits, dev(G["sampled"].T.copy(), torch.int32).reshape(-1), coef, 0.0)
    loss = float(torch.dot(coef, nll) / float(G["mask"].sum()))
    assert abs(loss - float(G["pg_loss"])) < 1e-3 * max(1.0, abs(float(G["pg_loss"])))       # north_star: fp losses within 1e-3
