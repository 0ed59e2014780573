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


def test_multilabel_and_get_captions_match_the_reference_functions():
    """get_multilabel (reinforce_multitask_e2e_attribute_loss.py:874-893) and get_captions (:871-872), pinned to outputs of the reference's own
    functions executed in the build container (tools/make_fixtures.py): real MSVD captions against an attribute vocabulary with a duplicate
    and an out-of-vocabulary entry, repeated words, double spaces, a multi-word "attribute" (never matches a split word), a video with no hit."""
    from s2vt_amd import data
    G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hostglue.json")))
    assert len(G["multilabel_cases"]) == 2
    for c in G["multilabel_cases"]:
        got = hg.get_multilabel(c["vid_sentence"], c["vocabulary"])
        assert set(got) == set(c["labels"])
        for vid, lab in c["labels"].items():
            assert got[vid].tolist() == lab, vid
    dup = G["multilabel_cases"][0]
    first = next(iter(dup["labels"].values()))
    i0, i1 = [i for i, w in enumerate(dup["vocabulary"]) if w == "man"]
    assert first[i0] == first[i1] == 1                                   # both positions of a duplicated attribute word are set
    gc = G["get_captions_cases"]
    idx = data.CaptionIndex([tuple(p) for p in gc["captions"]])
    for vid, out in gc["queries"].items():
        assert (idx.get_captions(vid) if vid in idx.by_video else []) == out


def test_multilabel_metrics_and_file_readers_match_the_reference_functions(tmp_path):
    """get_metrics (reinforce_multitask_e2e_attribute_loss.py:700-717: a score exactly at the threshold is a predicted positive) and
    read_sent_vocab_file (:852-869), pinned to the reference's own functions; the summary expressions of its test loop (:1066-1073)."""
    G2 = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hostglue.json")))
    for c in G2["metric_cases"]:
        out = hg.get_metrics(c["scores"], c["labels"], c["threshold"])
        assert list(out) == c["out"]
        assert out[0] + out[3] == out[4] and out[1] + out[2] == out[5]
    tp, tn, fp, fn, cp, cn = G2["metric_cases"][2]["out"]
    sm = hg.multilabel_summary(tp, tn, fp, fn, cp, cn)
    assert abs(sm["sensitivity"] - tp / cp) < 1e-12 and abs(sm["f1_score"] - 2 * tp / (2 * tp + fp + fn)) < 1e-12
    assert abs(sm["harmmean"] - 2.0 / (1.0 / sm["sensitivity"] + 1.0 / sm["specificity"])) < 1e-12
    assert hg.multilabel_summary(0, 0, 0, 0, 0, 0) == {"sensitivity": 0, "specificity": 0, "harmmean": 0, "precision": 0, "f1_score": 0}
    import pytest
    with pytest.raises(ZeroDivisionError):                               # as the reference's expression: no true positive among counted labels
        hg.multilabel_summary(0, 5, 0, 0, 0, 5)
    c = G2["sent_vocab_case"]
    (tmp_path / "s.txt").write_text(c["sent_text"]); (tmp_path / "v.txt").write_text(c["vocab_text"])
    vs, vb, n = hg.read_sent_vocab_file(str(tmp_path / "s.txt"), str(tmp_path / "v.txt"))
    assert vs == c["vid_sent"] and vb == c["vocab"] and n == c["label_num"]
