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
