#!/usr/bin/env python3
"""Generate tests/golden/hostglue.json by EXECUTING the reference's own TF-free host helpers.

Runs ONLY in the build container (needs /root/reference); the GPU box never sees the
reference.  The helpers are Python-2 source, so the relevant function bodies are read from the
reference files at run time, the py2-only `print` statements are dropped, and the text is
exec'd in a scratch namespace with numpy.  Nothing of the reference's source is written to
the fixture: it holds inputs (captions, id arrays) and the outputs the reference code returned.

Helpers executed:
  tf_s2vt.py                : preProBuildWordVocab, sentence_padding_toix
  cider_evaluation.py       : decode_captions, decode_captions_masks
  reinforce_multitask_e2e_attribute_loss.py : get_multilabel (bag-of-words attribute labels, :874-893), get_captions (:871-872)
  e2e_tf_s2vt.py            : get_video_feature_caption_pair (which frames of a video feed the CNN, :376-412) on a scratch tree of empty files
  tf_s2vt.py                : get_video_feature_caption_pair (the feature text-file parser, :324-345) on a synthetic file
  reinforce_multitask_e2e_attribute_loss.py : get_metrics (:700-717), read_sent_vocab_file (:852-869)
"""
import json
import os
import re
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "hostglue.json")


def grab(path, name):
    src = open(os.path.join(REF, path)).read().expandtabs(8).split("\n")
    start = next(i for i, l in enumerate(src) if l.startswith("def %s(" % name))
    end = next((i for i in range(start + 1, len(src)) if src[i] and not src[i][0].isspace() and not src[i].startswith("#")),
               len(src))
    body = [l for l in src[start:end] if not re.match(r"\s*print\s", l)]
    return "\n".join(body)


def main():
    ns = {"np": np}
    for path, name in [("tf_s2vt.py", "preProBuildWordVocab"), ("tf_s2vt.py", "sentence_padding_toix"),
                       ("cider_evaluation.py", "decode_captions"), ("cider_evaluation.py", "decode_captions_masks")]:
        exec(compile(grab(path, name), path + ":" + name, "exec"), ns)

    vocab = [l.rstrip() for l in open(os.path.join(REF, "msvd_vocabulary1.txt"))]
    wordtoix, ixtoword = ns["preProBuildWordVocab"](vocab, word_count_threshold=0)

    lines = [l.rstrip("\n").split("\t") for l in open(os.path.join(REF, "msvd_sents_train_noval_lc_nopunc.txt"))]
    caps = [c for _, c in lines]
    rng = np.random.default_rng(7)
    pick = [caps[i] for i in rng.choice(len(caps), 40, replace=False)]
    longest = sorted(caps, key=lambda c: -len(c.split(" ")))[:4]
    special = ["a zzzunknownword is eating qqqnotaword", "A Man IS Running", "", "man", "a  double space"]
    captions = pick + longest + special

    cases = []
    for Tc in (20, 35, 5):
        ns["n_caption_lstm_step"] = Tc
        batch = list(captions)
        ids, mask = ns["sentence_padding_toix"](batch, wordtoix)
        cases.append({"n_caption_lstm_step": Tc, "ids": [[int(x) for x in r] for r in ids],
                      "mask": np.asarray(mask).astype(int).tolist()})

    # decode + PG masks on id arrays: real padded captions, random ids, rows w/o <eos>, <eos> first
    dec_inputs = []
    ns["n_caption_lstm_step"] = 20
    ids20, _ = ns["sentence_padding_toix"](list(captions), wordtoix)
    dec_inputs.append(np.asarray(ids20))
    r = rng.integers(0, 30, size=(24, 20))          # many <eos>=0 hits
    dec_inputs.append(r)
    dec_inputs.append(rng.integers(2, len(vocab) + 2, size=(6, 20)))  # never stops
    dec_inputs.append(np.array([[3, 4, 0, 0, 3], [3, 3, 3, 3, 3], [0, 4, 4, 4, 4]]))
    dec_inputs.append(np.array([5, 6, 0, 7]))       # 1-D input branch
    dec_cases = []
    for a in dec_inputs:
        masks, strings = ns["decode_captions_masks"](a, ixtoword)
        plain = ns["decode_captions"](a, ixtoword)
        dec_cases.append({"ids": a.tolist(), "masks": [[int(x) for x in m] for m in masks], "decoded": strings,
                          "decoded_plain": plain})

    # ---- attribute labels of the multitask scripts: the reference's own get_multilabel / get_captions (py2 dict.iteritems -> a dict subclass)
    from collections import defaultdict
    ns["defaultdict"] = defaultdict
    for name in ("get_multilabel", "get_captions"):
        exec(compile(grab("reinforce_multitask_e2e_attribute_loss.py", name), "reinforce_multitask_e2e_attribute_loss.py:" + name, "exec"), ns)

    class Py2Dict(dict):
        iteritems = dict.items
    by_vid = {}
    for vid, c in lines[:400]:
        by_vid.setdefault(vid, []).append(c)
    vids = list(by_vid)[:6]
    attr_vocab = ["man", "woman", "dog", "cat", "is", "playing", "guitar", "zzznotaword", "a", "man", "running", "cooking", "the"]   # incl. a duplicate and an OOV
    ml_cases = []
    for vv, vs in ((attr_vocab, {v: by_vid[v] for v in vids}),
                   (["x", "y", "x y"], {"v0": ["x x x", "y  y", "z"], "v1": ["x y"], "v2": ["z z"]})):
        lab = ns["get_multilabel"](Py2Dict(vs), vv)
        ml_cases.append({"vocabulary": vv, "vid_sentence": vs, "labels": {k: [int(x) for x in np.asarray(v).reshape(-1)] for k, v in lab.items()}})
    pairs = [(vid, c) for vid, c in lines[:60]]
    gc_cases = {"captions": [list(p_) for p_ in pairs], "queries": {v: ns["get_captions"](pairs, v) for v in (pairs[0][0], pairs[-1][0], "no-such-video")}}

    # ---- frame selection of the end-to-end scripts: the reference's own get_video_feature_caption_pair (e2e_tf_s2vt.py:376-412) on a scratch
    # tree of empty %06d.jpg files -- which frame numbers a video of a given length contributes
    import glob as _glob, tempfile
    ns.update({"glob": _glob, "os": os, "sys": sys, "video_train_sent_file": "", "video_path": "", "n_video_lstm_step": 5})
    exec(compile(grab("e2e_tf_s2vt.py", "get_video_feature_caption_pair"), "e2e_tf_s2vt.py:get_video_feature_caption_pair", "exec"), ns)
    frame_cases = []
    with tempfile.TemporaryDirectory() as td:
        counts = [1, 2, 3, 5, 6, 7, 9, 10, 11, 14, 23, 24, 30, 31, 100, 101, 300]
        sent = os.path.join(td, "sents.txt")
        with open(sent, "w") as f:
            for c in counts:
                os.makedirs(os.path.join(td, "frames", "vid%d" % c))
                for k in range(1, c + 1):
                    open(os.path.join(td, "frames", "vid%d" % c, "%06d.jpg" % k), "w").close()
                f.write("vid%d\ta caption of video %d\n" % (c, c))
                f.write("vid%d\ta second caption\n" % c)
        for n in (5, 10, 2):
            sents_out, vf = ns["get_video_feature_caption_pair"](sent, os.path.join(td, "frames"), n)
            frame_cases.append({"num_frame_per_video": n, "n_sents": int(len(sents_out)), "first_sent": [str(x) for x in sents_out[0]],
                                "ticks": {str(c): [int(os.path.basename(p_)[:6]) for p_ in vf["vid%d" % c]] for c in counts}})

    # ---- the feature text files: the reference's own parser (tf_s2vt.py:324-345) on a synthetic file -- video id = text before the first '_',
    # frames in FILE order (interleaved videos, frame numbers out of order), values as the strings TF then converts to float32
    ns.update({"video_train_feature_file": ""})
    exec(compile(grab("tf_s2vt.py", "get_video_feature_caption_pair"), "tf_s2vt.py:get_video_feature_caption_pair", "exec"), ns)
    frng = np.random.default_rng(11)
    feat_lines = []
    order_ = [("vid7", 3), ("vid12", 1), ("vid7", 1), ("vid12", 2), ("vid3", 1), ("vid7", 2), ("vid3", 3), ("vid12", 3), ("vid3", 2)]
    fmts = ["%.6f", "%.9g", "%r", "%.3e", "%g"]
    for li, (v, k) in enumerate(order_):
        vals = np.abs(frng.standard_normal(6) * 0.5)
        vals[li % 6] = [0.0, 1.0, 0.1, 1e-8, 123456.789, 3.4028234e38, 1e-45, 0.30000001192092896, 2.5][li]
        feat_lines.append("%s_frame_%d," % (v, k) + ",".join(fmts[(li + j) % 5] % float(x) for j, x in enumerate(vals)))
    feat_text = "\n".join(feat_lines) + "\n"
    with tempfile.TemporaryDirectory() as td:
        ff, sf = os.path.join(td, "feats.txt"), os.path.join(td, "sents.txt")
        open(ff, "w").write(feat_text)
        open(sf, "w").write("vid7\ta b c\nvid3\td e\n")
        sents_f, feats_f = ns["get_video_feature_caption_pair"](sf, ff)
    feature_case = {"file_text": feat_text, "sents": [[str(a), str(b)] for a, b in sents_f],
                    # what feeding the parsed strings to a float32 placeholder gives (numpy: string -> double -> float32), bit patterns
                    "features_f32_bits": {v: np.asarray(rows, dtype=np.float32).view(np.uint32).tolist() for v, rows in feats_f.items()},
                    "video_order": list(feats_f.keys())}

    # ---- multilabel evaluation: the reference's own get_metrics (reinforce_multitask_e2e_attribute_loss.py:700-717; globals nums_label, threshold)
    # and read_sent_vocab_file (:852-869)
    ns.update({"xrange": range, "nums_label": 7, "threshold": 0.5})
    exec(compile(grab("reinforce_multitask_e2e_attribute_loss.py", "get_metrics"), "reinforce_multitask_e2e_attribute_loss.py:get_metrics", "exec"), ns)
    exec(compile(grab("reinforce_multitask_e2e_attribute_loss.py", "read_sent_vocab_file"), "reinforce_multitask_e2e_attribute_loss.py:read_sent_vocab_file", "exec"), ns)
    metric_cases = []
    for nv in (1, 5, 12):
        sc = np.round(frng.random((nv, 7)), 2); sc[0, 0] = 0.5; sc[-1, -1] = 0.49999
        lb = (frng.random((nv, 7)) < 0.4).astype(np.int64)
        out_m = ns["get_metrics"](sc, lb, nv)
        metric_cases.append({"scores": sc.tolist(), "labels": lb.tolist(), "threshold": 0.5, "out": [int(x) for x in out_m]})
    with tempfile.TemporaryDirectory() as td:
        sf, vf = os.path.join(td, "s.txt"), os.path.join(td, "v.txt")
        stxt = "vid1\ta man is running\nvid2\ta dog\nvid1\tanother  caption \n"
        vtxt = "man\ndog\n running \n"
        open(sf, "w").write(stxt); open(vf, "w").write(vtxt)
        vs_, vb_, ln_ = ns["read_sent_vocab_file"](sf, vf)
    sent_vocab_case = {"sent_text": stxt, "vocab_text": vtxt, "vid_sent": vs_, "vocab": vb_, "label_num": int(ln_)}

    probe = ["<eos>", "<bos>", "<en_unk>", "a", "man", "is", "the", vocab[-1]]
    # vocabulary subset needed to replay the cases without shipping the reference's vocabulary file
    used_ids = set()
    for c in dec_cases:
        used_ids.update(int(x) for x in np.asarray(c["ids"]).reshape(-1))
    used_words = set(w for cap in captions for w in cap.lower().split(" ") if w in wordtoix)
    subset = {ixtoword[i]: int(i) for i in used_ids}
    subset.update({w: int(wordtoix[w]) for w in used_words})
    subset.update({w: int(wordtoix[w]) for w in probe})
    toy_w2i, toy_i2w = ns["preProBuildWordVocab"](["<en_unk>", "b", "c"], word_count_threshold=0)
    out = {
        "generator": "tools/make_fixtures.py (exec of reference helpers under py3)",
        "vocab_size": len(wordtoix),
        "wordtoix_probe": {w: int(wordtoix[w]) for w in probe},
        "vocab_subset": subset,
        "toy_vocab": {"wordtoix": {k: int(v) for k, v in toy_w2i.items()},
                      "ixtoword": {str(k): v for k, v in toy_i2w.items()}},
        "captions": captions,
        "padding_cases": cases,
        "decode_cases": dec_cases,
        "multilabel_cases": ml_cases,
        "get_captions_cases": gc_cases,
        "frame_tick_cases": frame_cases,
        "feature_file_case": feature_case,
        "metric_cases": metric_cases,
        "sent_vocab_case": sent_vocab_case,
    }
    with open(OUT, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", os.path.normpath(OUT), os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    sys.exit(main())


def msvd_slice(out_path):
    """tests/golden/msvd_slice.json: a slice of the reference's own DATA files -- the captions of the first 40
    training videos (msvd_sents_train_noval_lc_nopunc.txt) and the part of msvd_vocabulary1.txt they use (with
    ~5% of those words dropped so that the slice has out-of-vocabulary reference words)."""
    refs, order = {}, []
    for line in open(os.path.join(REF, "msvd_sents_train_noval_lc_nopunc.txt")):
        vid, sent = line.rstrip("\n").split("\t")
        if vid not in refs:
            if len(order) == 40:
                continue
            refs[vid] = []
            order.append(vid)
        refs[vid].append(sent)
    vocab_all = [l.rstrip() for l in open(os.path.join(REF, "msvd_vocabulary1.txt"))]
    used = set(w for v in order for s in refs[v] for w in s.split())
    vocab = [w for i, w in enumerate(vocab_all) if w in used or i < 300]
    vocab = [w for i, w in enumerate(vocab) if not (i % 19 == 7 and w != "<en_unk>")]
    json.dump({"source": "msvd_sents_train_noval_lc_nopunc.txt (first 40 videos) + msvd_vocabulary1.txt (subset), made by tools/make_fixtures.py",
               "vocab": vocab, "refs_by_video": [refs[v] for v in order]}, open(out_path, "w"))


if __name__ == "__main__" and os.path.isdir(REF):
    msvd_slice(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "msvd_slice.json"))
