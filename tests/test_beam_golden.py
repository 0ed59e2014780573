"""Beam search pinned to the reference's own bookkeeping (tests/golden/beam_search.json, made by
tools/make_beam_fixtures.py from the reference's Caption / TopN classes, beam_search.py:6-80, and the expansion loop of
final_beam_search.py:226-294).  CPU part: BestK reproduces TopN's push / extract(sort=True) traces, ties included.
GPU part: BeamSearchGenerator returns the fixture's captions, log-probabilities and scores."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "beam_search.json")


def test_bestk_matches_reference_topn_traces():
    from s2vt_amd.beam_generator import BestK, Hypothesis
    gold = json.load(open(GOLD))
    assert len(gold["topn_traces"]) >= 5
    for tr in gold["topn_traces"]:
        t = BestK(tr["n"])
        for i, s in enumerate(tr["scores"]):
            t.push(Hypothesis([i], 0, s, s))
        assert t.size() == min(tr["n"], len(tr["scores"]))
        got = [[h.score, h.sentence[0]] for h in t.best_first()]
        assert got == tr["extract_sorted"], tr["n"]
        t.clear()
        assert t.size() == 0


@pytest.mark.gpu
def test_beam_generator_matches_reference_fixture(gpu, oracle):
    import torch
    from s2vt_amd import model as M
    from s2vt_amd.beam_generator import BeamSearchGenerator
    gold = json.load(open(GOLD))
    dm = gold["dims"]
    d = oracle.Dims(label_dim=0, **dm)
    models = {}
    finished = 0
    for c in gold["cases"]:
        ps = c["param_seed"]
        if ps not in models:
            p = oracle.init_params(d, seed=ps)
            p["embed_word_W"] *= np.float32(gold["scales"]["embed_word_W"])
            p["lstm1_W"] *= np.float32(gold["scales"]["lstm_W"]); p["lstm2_W"] *= np.float32(gold["scales"]["lstm_W"])
            p["Wemb"] *= np.float32(gold["scales"]["Wemb"])
            p["embed_word_b"][0] += np.float32(gold["eos_bias"][str(ps)])
            mdl = M.Video_Caption_Generator(dm["dim_image"], dm["n_words"], dm["word_dim"], dm["lstm_dim"], 1, 0,
                                            dm["n_video_lstm_step"], dm["n_caption_lstm_step"])
            mdl.store.load(p)
            models[ps] = mdl
        mdl = models[ps]
        video = np.asarray(c["video"], np.float32).reshape(1, dm["n_video_lstm_step"], dm["dim_image"])
        s, lp, sc = BeamSearchGenerator(mdl, c["beam_size"], c["length_normalization_factor"]).generate(video)
        tag = (ps, c["beam_size"], c["length_normalization_factor"])
        assert s == c["sentence"], tag
        assert abs(lp - c["logprob"]) <= 2e-4 * max(1.0, abs(c["logprob"])), tag
        assert abs(sc - c["score"]) <= 2e-4 * max(1.0, abs(c["score"])), tag
        finished += s[-1] == 0
    assert 0 < finished < len(gold["cases"])               # both exits of the loop are covered
