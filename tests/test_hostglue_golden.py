"""Host glue vs the golden pairs produced by EXECUTING the reference's own helpers
(tools/make_fixtures.py; tf_s2vt.py:347-401, cider_evaluation.py:122-172)."""
import json
import os

import numpy as np

import s2vt_amd
from s2vt_amd import hostglue as hg

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "hostglue.json")))


def test_vocab_builder():
    w2i, i2w = hg.preProBuildWordVocab(["<en_unk>", "b", "c"])
    assert w2i == G["toy_vocab"]["wordtoix"]
    assert {str(k): v for k, v in i2w.items()} == G["toy_vocab"]["ixtoword"]


def test_sentence_padding_toix():
    w2i = dict(G["vocab_subset"])
    for case in G["padding_cases"]:
        caps = list(G["captions"])
        ids, mask = hg.sentence_padding_toix(caps, w2i, case["n_caption_lstm_step"])
        assert ids == case["ids"]
        assert np.array_equal(np.asarray(mask).astype(int), np.asarray(case["mask"]))
        assert caps == G["captions"]            # caller's list untouched


def test_decode_captions_and_masks():
    i2w = {v: k for k, v in G["vocab_subset"].items()}
    for case in G["decode_cases"]:
        ids = np.asarray(case["ids"])
        masks, dec = hg.decode_captions_masks(ids, i2w)
        assert masks == case["masks"] and dec == case["decoded"]
        assert hg.decode_captions(ids, i2w) == case["decoded_plain"]
        m = hg.masks_from_ids(ids if ids.ndim == 2 else ids[None])
        assert np.array_equal(m.astype(int), np.asarray(case["masks"]))


def test_tiling_is_sample_major():
    x = np.arange(6).reshape(3, 2)
    t = hg.tile_k(x, 4)
    for k in range(4):
        for j in range(3):
            assert np.array_equal(t[k * 3 + j], x[j])    # row k*B + j = copy k of video j
    assert np.array_equal(hg.tile_baseline([1., 2., 3.], 2), [1, 2, 3, 1, 2, 3])


def test_multilabel():
    lab = hg.get_multilabel({"v": ["a man is running", "a dog"]}, ["man", "cat", "dog"])
    assert lab["v"].tolist() == [1, 0, 1]
