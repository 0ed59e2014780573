"""Host-side glue of the REINFORCE step: vocabulary, padding, decode + PG mask, K-tiling.

These mirror the reference's pure-Python helpers (same names, same argument meaning) so a
`train()` body written against the reference reads the same here:

* ``preProBuildWordVocab``   -- tf_s2vt.py:347-368  (``<eos>``=0, ``<bos>``=1, vocab word i -> i+2)
* ``sentence_padding_toix``  -- tf_s2vt.py:371-401 / reinforcement_multisampling_tf_s2vt.py:568-598
* ``decode_captions``        -- cider_evaluation.py:122-143
* ``decode_captions_masks``  -- cider_evaluation.py:145-172 (mask = 1 up to AND including first <eos>)
* ``tile_k`` / ``tile_baseline`` -- reinforcement_multisampling_tf_s2vt.py:764-782, 790-795

The array forms (``masks_from_ids``) are what the trainer uses on the hot path: O(N*Tc) numpy,
no Python loop per token.  Pinned by tests/golden/hostglue.json, which was produced by
executing the reference's own helper source in the build container (tools/make_fixtures.py).
"""
from __future__ import annotations

import numpy as np

EOS, BOS = 0, 1


def preProBuildWordVocab(vocabulary, word_count_threshold=0):
    ixtoword = {1: "<bos>", 0: "<eos>"}
    wordtoix = {"<bos>": 1, "<eos>": 0}
    for idx, w in enumerate(vocabulary):
        wordtoix[w] = idx + 2
        ixtoword[idx + 2] = w
    return wordtoix, ixtoword


def sentence_padding_toix(captions_batch, wordtoix, n_caption_lstm_step):
    """Returns (ids [N, Tc] list of lists, mask [N, Tc] float array).

    Captions shorter than Tc get ``<eos>`` appended up to Tc with mask 1 on the words and the
    FIRST ``<eos>``; longer ones are truncated to Tc-1 words + ``<eos>`` (mask all ones).
    Out-of-vocabulary words map to ``<en_unk>``.  The reference mutates ``captions_batch`` in
    place; this version leaves the caller's list alone and returns the same values.
    """
    Tc = n_caption_lstm_step
    masks = np.ones((len(captions_batch), Tc))
    ids = []
    unk = wordtoix["<en_unk>"]
    for n, cap in enumerate(captions_batch):
        words = cap.lower().split(" ")
        if len(words) < Tc:
            masks[n, len(words) + 1:] = 0
            words = words + ["<eos>"] * (Tc - len(words))
        else:
            words = words[:Tc - 1] + ["<eos>"]
        ids.append([wordtoix.get(w, unk) for w in words])
    return ids, masks


def decode_captions(captions, idx_to_word):
    captions = np.asarray(captions)
    if captions.ndim == 1:
        captions = captions[None]
    out = []
    for row in captions:
        words = []
        for t in row:
            w = idx_to_word[int(t)]
            if w == "<eos>":
                break
            words.append(w)
        out.append(" ".join(words))
    return out


def masks_from_ids(ids, eos: int = EOS):
    """[N,Tc] ids -> float32 mask, 1 up to and including the first <eos>, 0 after."""
    ids = np.asarray(ids)
    is_eos = ids == eos
    before = np.cumsum(is_eos, axis=1) - is_eos
    return (before == 0).astype(np.float32)


def decode_captions_masks(captions, idx_to_word):
    captions = np.asarray(captions)
    if captions.ndim == 1:
        captions = captions[None]
    return masks_from_ids(captions).astype(np.int64).tolist(), decode_captions(captions, idx_to_word)


def tile_k(x, K: int):
    """Sample-major tiling: row k*B + j = copy k of row j (what the reference builds with nested
    loops over `features_batch8`)."""
    x = np.asarray(x)
    return np.tile(x, (K,) + (1,) * (x.ndim - 1))


def tile_baseline(b, K: int):
    return np.tile(np.asarray(b).reshape(-1), K)


def get_multilabel(vid_sentence, vocabulary):
    """Bag-of-words attribute labels (reinforce_multitask_e2e_attribute_loss.py:874-893):
    label[v] = 1 iff attribute word v occurs in any caption of the video."""
    index = {}
    for i, w in enumerate(vocabulary):                 # (the reference scans the vocabulary per word: EVERY position holding the word is set,
        index.setdefault(w, []).append(i)              #  should the attribute vocabulary list a word twice -- tests/golden/hostglue.json)
    out = {}
    for vid, sents in vid_sentence.items():
        lab = np.zeros(len(vocabulary), np.int64)
        for s in sents:
            for w in s.split():
                for i in index.get(w, ()):
                    lab[i] = 1
        out[vid] = lab
    return out


def read_sent_vocab_file(sent_file, vocab_file):
    """(video -> list of its sentences, attribute vocabulary, label_num): reinforce_multitask_e2e_attribute_loss.py:852-869 (lines stripped,
    sentences grouped per video in file order)."""
    vid_sent, vocab = {}, []
    with open(sent_file) as f:
        for line in f:
            a = line.strip().split("\t")
            vid_sent.setdefault(a[0], []).append(a[1])
    with open(vocab_file) as f:
        for line in f:
            vocab.append(line.strip())
    return vid_sent, vocab, len(vocab)


def get_metrics(scores, labels, threshold=0.5):
    """(true_positive, true_negative, false_positive, false_negative, count_pos, count_neg) of the multilabel evaluation
    (reinforce_multitask_e2e_attribute_loss.py:700-717): a label >= threshold is a positive, a score >= threshold a predicted positive.
    scores / labels [num_videos, label_dim] (the scores of evaluate_multilabel, the bag-of-words labels of get_multilabel)."""
    s = np.asarray(scores) >= threshold
    pos = np.asarray(labels) >= threshold
    return (int((s & pos).sum()), int((~s & ~pos).sum()), int((s & ~pos).sum()), int((~s & pos).sum()), int(pos.sum()), int((~pos).sum()))


def multilabel_summary(true_positive, true_negative, false_positive, false_negative, count_pos, count_neg):
    """sensitivity, specificity, their harmonic mean, precision and F1 as the reference's test loop forms them (:1066-1073), zero guards included
    (like the reference, the harmonic mean divides by the true positives / negatives: no true positive or no true negative raises ZeroDivisionError)."""
    sensitivity = true_positive / float(count_pos) if count_pos > 0 else 0
    specificity = true_negative / float(count_neg) if count_neg > 0 else 0
    harmmean = 2.0 / (count_pos / float(true_positive) + count_neg / float(true_negative)) if (count_pos + count_neg) > 0 else 0
    precision = (true_positive / float(true_positive + false_positive)) if true_positive > 0 else 0
    f1_score = 2.0 * true_positive / float(2 * true_positive + false_positive + false_negative) if true_positive > 0 else 0
    return {"sensitivity": sensitivity, "specificity": specificity, "harmmean": harmmean, "precision": precision, "f1_score": f1_score}
