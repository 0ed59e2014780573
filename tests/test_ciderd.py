"""The C++ id-based CIDEr-D (libs2vt_host.so) against the plain-Python restatement on word strings, on a slice
of the reference's own MSVD sentences (committed as a small fixture) incl. OOV words, empty and over-long
candidates."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _corpus():
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "msvd_slice.json")))
    return fx["vocab"], fx["refs_by_video"]


def test_ciderd_ids_match_string_restatement():
    import s2vt_amd
    from s2vt_amd import hostglue, reward
    from oracle import ciderd_py
    vocab, refs = _corpus()
    wordtoix, ixtoword = hostglue.preProBuildWordVocab(vocab)
    sc = reward.CiderD(refs, wordtoix, n_threads=2)
    py = ciderd_py.CiderD(refs)
    rng = np.random.default_rng(0)
    Tc, N = 12, 64
    ids = np.zeros((N, Tc), np.int32)
    vid = rng.integers(0, len(refs), N).astype(np.int32)
    for n in range(N):
        words = refs[vid[n]][rng.integers(len(refs[vid[n]]))].split()        # start from a real reference ...
        toks = [wordtoix.get(w, 2) for w in words][:Tc]
        for i in range(len(toks)):                                             # ... and perturb it
            if rng.random() < 0.25:
                toks[i] = int(rng.integers(1, len(wordtoix)))
        ids[n, :len(toks)] = toks
    ids[0] = 0                                   # empty caption
    ids[1] = 7                                   # never stops: all Tc tokens are words
    got = sc.score_ids(ids, vid)
    for n in range(N):
        s = hostglue.decode_captions(ids[n], ixtoword)[0]
        want = py.score(s, int(vid[n]))
        assert abs(got[n] - want) <= 1e-5 * max(1.0, abs(want)), (n, s, got[n], want)
    assert got.max() > 1.0 and got[0] == 0.0     # the slice produces real scores; the empty caption scores 0


def test_ciderd_exact_reference_scores_high_and_bad_args():
    import s2vt_amd
    from s2vt_amd import hostglue, reward
    vocab, refs = _corpus()
    wordtoix, _ = hostglue.preProBuildWordVocab(vocab)
    sc = reward.CiderD(refs, wordtoix)
    words = refs[3][0].split()
    ids = np.zeros((2, 20), np.int32)
    ids[0, :len(words)] = [wordtoix.get(w, 2) for w in words]
    ids[1, :3] = [wordtoix[w] for w in vocab[50:53]]
    r = sc.score_ids(ids, [3, 3])
    assert r[0] > 3 * max(r[1], 0.1)
    with pytest.raises(ValueError):
        sc.score_ids(ids, [3, len(refs)])
