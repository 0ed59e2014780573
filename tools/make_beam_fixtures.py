#!/usr/bin/env python3
"""Generate tests/golden/beam_search.json by EXECUTING the reference's own beam bookkeeping.

Runs ONLY in the build container (needs /root/reference).  `beam_search.py` parses under Python 3; its module-level
`import tensorflow` is the only obstacle, so the file is read at run time, that import line is dropped and the text is
exec'd in a scratch namespace: `Caption` and `TopN` below ARE the reference's classes (beam_search.py:6-80).  The
expansion loop of final_beam_search.py:226-294 (Python-2 source inside a TF graph builder, not importable) is driven
here statement by statement on those classes, with `sess.run(beam_probability)` answered by the CPU oracle
(oracle/s2vt_oracle.py: one LSTM1 + LSTM2 + vocab step at B = 1, probabilities exp(l)/sum(exp(l)) and top-k as
final_beam_search.py:218-220).  Nothing of the reference's source is written to the fixture: it holds inputs (seeds,
feature blocks, beam sizes), the captions / log-probabilities / scores that came out, and push/extract traces of TopN.
"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden", "beam_search.json")

# Random-init weights give near-uniform next-word distributions and near-zero states; scaled up, beams really compete
# and the state matters.  <eos> bias per parameter seed: a mix of captions that finish early and ones that run to Tc.
W_SCALE, LSTM_SCALE, WEMB_SCALE = 30.0, 6.0, 20.0
EOS_BIAS = {3: 2.0, 4: 4.0}
DIMS = dict(dim_image=24, n_words=60, word_dim=12, lstm_dim=20, n_video_lstm_step=3, n_caption_lstm_step=7)


def reference_classes():
    src = open(os.path.join(REF, "beam_search.py")).read()
    src = "\n".join(l for l in src.split("\n") if not l.startswith("import tensorflow"))
    ns = {}
    exec(compile(src, "beam_search.py", "exec"), ns)
    return ns["Caption"], ns["TopN"]


def make_params(orc, seed):
    d = orc.Dims(label_dim=0, **DIMS)
    p = orc.init_params(d, seed=seed)
    p["embed_word_W"] *= np.float32(W_SCALE)
    p["lstm1_W"] *= np.float32(LSTM_SCALE)
    p["lstm2_W"] *= np.float32(LSTM_SCALE)
    p["Wemb"] *= np.float32(WEMB_SCALE)
    p["embed_word_b"][0] += np.float32(EOS_BIAS[seed])
    return d, p


def beam_probability(orc, p, state1, state2, word, beam_size):
    """final_beam_search.py:203-224 at B = 1: (top-k word ids, their probabilities, new state2, new state1)."""
    c1, h1 = state1
    c2, h2 = state2
    c1, h1, o1, _, _ = orc.lstm1_step(p, None, c1, h1)
    c2, h2, o2, _, _ = orc.lstm2_step(p, o1, np.asarray([word], np.int32), c2, h2)
    logits = orc.xw_plus_b(o2, p["embed_word_W"], p["embed_word_b"])[0].astype(np.float64)
    e = np.exp(logits)
    probs = e / e.sum()
    order = np.argsort(-probs, kind="stable")[:beam_size]            # tf.nn.top_k: descending, lowest index first on ties
    return order.astype(int), probs[order], (c2, h2), (c1, h1)


def build_generator(Caption, TopN, orc, d, p, video, beam_size, length_normalization_factor):
    """The body of final_beam_search.py:226-294 on the reference's TopN / Caption."""
    c1, h1, c2, h2 = orc.encode(p, orc.frame_embed(p, video))
    initial_state1, initial_state2 = (c1, h1), (c2, h2)
    captions = TopN(beam_size * beam_size)
    final_captions = TopN(beam_size)
    initial_word = 1
    word_index, probs, state2, state1 = beam_probability(orc, p, initial_state1, initial_state2, initial_word, beam_size)
    for beam in range(beam_size):
        captions.push(Caption(sentence=[int(word_index[beam])], img_state=state1, language_state=state2,
                              logprob=math.log(probs[beam]), score=math.log(probs[beam])))
    exclude_num = 0
    for i in range(1, d.n_caption_lstm_step):
        mid_captions = captions.extract(sort=True)[:beam_size]
        captions.reset()
        for mid_caption in mid_captions:
            word_index, probs, state2, state1 = beam_probability(orc, p, mid_caption.img_state, mid_caption.language_state,
                                                                 mid_caption.sentence[-1], beam_size)
            for beam in range(beam_size - exclude_num):
                sentence = mid_caption.sentence + [int(word_index[beam])]
                logprob = mid_caption.logprob + math.log(probs[beam])
                score = logprob
                if word_index[beam] == 0:
                    if length_normalization_factor > 0:
                        score /= len(sentence) ** length_normalization_factor
                    final_captions.push(Caption(sentence, state1, state2, logprob, score))
                    exclude_num += 1
                else:
                    captions.push(Caption(sentence, state1, state2, logprob, score))
        if exclude_num == beam_size:
            break
    if not final_captions.size():
        final_captions = captions
    final_cap = final_captions.extract(sort=True)[0]
    return final_cap.sentence, final_cap.logprob, final_cap.score


def topn_traces(Caption, TopN):
    """Push sequences (with exact score ties) -> what extract(sort=True) returns, as (score, arrival index) pairs."""
    rng = np.random.default_rng(11)
    traces = []
    for n, count, levels in [(3, 10, 4), (9, 40, 6), (1, 5, 2), (4, 3, 3), (5, 30, 1000)]:
        scores = (rng.integers(0, levels, count) / 4.0 - 2.0).tolist()
        t = TopN(n)
        for i, s in enumerate(scores):
            t.push(Caption([i], None, None, s, s))
        out = t.extract(sort=True)
        traces.append({"n": n, "scores": scores, "extract_sorted": [[c.score, c.sentence[0]] for c in out]})
    return traces


def main():
    from oracle import s2vt_oracle as orc
    Caption, TopN = reference_classes()
    cases = []
    rng = np.random.default_rng(5)
    for pseed in (3, 4):
        d, p = make_params(orc, pseed)
        for v in range(3):
            video = np.abs(rng.standard_normal((1, d.n_video_lstm_step, d.dim_image))).astype(np.float32)
            for beam, lnf in [(1, 0.0), (2, 0.0), (3, 0.0), (3, 0.5), (5, 0.5), (5, 1.0)]:
                s, lp, sc = build_generator(Caption, TopN, orc, d, p, video, beam, lnf)
                cases.append({"param_seed": pseed, "video": video.reshape(-1).tolist(), "beam_size": beam,
                              "length_normalization_factor": lnf, "sentence": [int(x) for x in s], "logprob": lp, "score": sc})
    out = {"generator": "tools/make_beam_fixtures.py (reference beam_search.py Caption/TopN executed; loop of "
                        "final_beam_search.py:226-294 driven on them with oracle step outputs)",
           "dims": DIMS, "scales": {"embed_word_W": W_SCALE, "lstm_W": LSTM_SCALE, "Wemb": WEMB_SCALE},
           "eos_bias": {str(k): v for k, v in EOS_BIAS.items()}, "cases": cases, "topn_traces": topn_traces(Caption, TopN)}
    json.dump(out, open(OUT, "w"))
    print("wrote", OUT, len(cases), "cases;", sum(1 for c in cases if c["sentence"][-1] == 0), "end in <eos>")


if __name__ == "__main__":
    main()
