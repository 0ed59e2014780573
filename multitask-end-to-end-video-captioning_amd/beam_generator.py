"""Beam-search generator: build_generator(beam_size, length_normalization_factor) of final_beam_search.py
:202-294 (its bookkeeping: the TopN heaps of Caption objects of beam_search.py:6-80).

The reference runs ONE sess.run per live beam per step (B = 1 graphs, states fed back through the host).  Here
a step advances ALL live beams in one batch on the device: LSTM1 once (its state never depends on a word, so it
is shared by every beam), LSTM2 + vocab logits at M = beams, log-softmax normaliser from the library's softmax
kernel; only the top-k selection and the caption bookkeeping (a few dozen scalars) run on the host, exactly as the
reference orders them -- including its quirks: candidates that end in <eos> leave the beam for good (exclude_num),
the score of a finished caption is logprob / len**factor, unfinished ones compete on raw logprob.
"""
from __future__ import annotations

import heapq

import numpy as np
import torch

from . import ops


class Hypothesis:
    """A partial caption: token ids so far, the LSTM2 state row it continues from, log-probability, ranking score.
    Ordered by score alone, like the reference's Caption (beam_search.py:24-42)."""
    __slots__ = ("sentence", "row", "logprob", "score")

    def __init__(self, sentence, row, logprob, score):
        self.sentence, self.row, self.logprob, self.score = sentence, row, logprob, score

    def __lt__(self, other):
        return self.score < other.score

    def __eq__(self, other):
        return self.score == other.score

    __hash__ = None


class BestK:
    """The k best-scoring hypotheses seen so far, with the reference's TopN semantics (beam_search.py:44-80) down to
    the order among EQUAL scores: a binary min-heap of at most k items (push while not full, push-then-pop-smallest
    once full), read out by a stable descending sort of the heap array.  Pinned by the push / extract traces in
    tests/golden/beam_search.json, which were produced by the reference's own class."""

    def __init__(self, k):
        self.k, self._heap = k, []

    def size(self):
        return len(self._heap)

    def push(self, h):
        if len(self._heap) < self.k:
            heapq.heappush(self._heap, h)
        else:
            heapq.heappushpop(self._heap, h)

    def best_first(self):
        out = list(self._heap)
        out.sort(reverse=True)
        return out

    def clear(self):
        self._heap = []


class BeamSearchGenerator:
    def __init__(self, model, beam_size=3, length_normalization_factor=0.0):
        self.m, self.beam_size, self.lnf = model, beam_size, length_normalization_factor

    # ---- device steps (all arithmetic in libs2vt_hip.so)
    def _encode(self, video):
        m = self.m
        p, H, E, Tv = m.store.p, m.lstm_dim, m.word_dim, m.n_video_lstm_step
        emb = ops.frame_embed_fwd(m.dims, m.store.params, video).view(1, Tv, E)
        z = torch.zeros(1, H, device=m.device)
        c1, h1, c2, h2 = z, z, z, z
        for t in range(Tv):                                  # final_beam_search.py:243-253
            c1, h1, _, _ = ops.lstm_cell_fwd(ops.operand(emb[:, t].contiguous()), None, h1, c1, p["lstm1_W"], p["lstm1_b"], 1)
            c2, h2, _, _ = ops.lstm_cell_fwd(ops.operand(h1), ops.operand(None, k=E), h2, c2, p["lstm2_W"], p["lstm2_b"], 1)
        return c1, h1, c2, h2

    def _step(self, c1, h1, c2, h2, words):
        """One decode step (beam_probability, :203-224) for len(words) beams: LSTM1 on its shared state (1 row),
        LSTM2 per beam; returns new states and (top-k word ids, their log-probs) per beam."""
        m = self.m
        p, E = m.store.p, m.word_dim
        n = len(words)
        c1, h1, _, _ = ops.lstm_cell_fwd(ops.operand(None, k=E), None, h1, c1, p["lstm1_W"], p["lstm1_b"], 1)
        idx = torch.as_tensor(words, dtype=torch.int32, device=m.device)
        c2, h2, _, _ = ops.lstm_cell_fwd(ops.operand(h1, rowmod=1), ops.operand(p["Wemb"], rowidx=idx), h2, c2, p["lstm2_W"], p["lstm2_b"], n)
        logits = ops.gemm([ops.operand(h2)], p["embed_word_W"], p["embed_word_b"], M=n)
        top_l, top_i = torch.topk(logits, self.beam_size, dim=1)                       # selection only
        zero_t = torch.zeros(n, dtype=torch.int32, device=m.device)
        _, lp0 = ops.softmax_nll_fwd_bwd(logits.clone(), zero_t, torch.zeros(n, device=m.device), 0.0)   # lp0 = l[0] - lse
        lse = logits[:, 0] - lp0
        return c1, h1, c2, h2, top_i.cpu().numpy(), (top_l - lse[:, None]).cpu().numpy()

    def generate_unshifted_softmax(self, video):
        """build_generator with the word choice exactly as tf_s2vt.py:208-209 writes it -- argmax of exp(l) / sum(exp(l))
        evaluated in fp32 WITHOUT a max shift -- instead of argmax(l): they differ when a logit overflows exp (>= 88.72:
        inf / inf = NaN, the choice becomes <eos> = 0) or when exp rounds two close logits to the same probability.
        Returns the Tc word ids (the reference's `break` at :212 never fires)."""
        m = self.m
        p, E = m.store.p, m.word_dim
        video = m._dev(video, torch.float32).view(1, m.n_video_lstm_step, m.dim_image)
        c1, h1, c2, h2 = self._encode(video)
        word = torch.ones(1, dtype=torch.int32, device=m.device)                       # <bos>
        ids = []
        for _ in range(m.n_caption_lstm_step):
            c1, h1, _, _ = ops.lstm_cell_fwd(ops.operand(None, k=E), None, h1, c1, p["lstm1_W"], p["lstm1_b"], 1)
            c2, h2, _, _ = ops.lstm_cell_fwd(ops.operand(h1), ops.operand(p["Wemb"], rowidx=word), h2, c2, p["lstm2_W"], p["lstm2_b"], 1)
            logits = ops.gemm([ops.operand(h2)], p["embed_word_W"], p["embed_word_b"], M=1)
            word, _ = ops.softmax_unshifted_argmax(logits)
            ids.append(word)
        return torch.cat(ids).cpu().numpy().astype(np.int64)

    def generate(self, video):
        """video [1, Tv, d] -> (sentence ids, logprob, score) of the best caption."""
        m, k = self.m, self.beam_size
        video = m._dev(video, torch.float32).view(1, m.n_video_lstm_step, m.dim_image)
        c1, h1, c2, h2 = self._encode(video)
        captions, final_captions = BestK(k * k), BestK(k)
        c1, h1, c2, h2, wi, lp = self._step(c1, h1, c2, h2, [1])                       # <bos>
        for b in range(k):
            captions.push(Hypothesis([int(wi[0, b])], 0, float(lp[0, b]), float(lp[0, b])))
        exclude = 0
        for _ in range(1, m.n_caption_lstm_step):
            mid = captions.best_first()[:k]
            captions.clear()
            if not mid:
                break
            rows = torch.as_tensor([cap.row for cap in mid], dtype=torch.long, device=m.device)
            c1, h1, c2n, h2n, wi, lp = self._step(c1, h1, c2[rows].contiguous(), h2[rows].contiguous(), [cap.sentence[-1] for cap in mid])
            c2, h2 = c2n, h2n
            for r, cap in enumerate(mid):
                for b in range(k - exclude):                                           # evaluated per caption, as the reference does
                    w = int(wi[r, b])
                    sentence = cap.sentence + [w]
                    logprob = cap.logprob + float(lp[r, b])
                    score = logprob
                    if w == 0:
                        if self.lnf > 0:
                            score /= len(sentence) ** self.lnf
                        final_captions.push(Hypothesis(sentence, r, logprob, score))
                        exclude += 1
                    else:
                        captions.push(Hypothesis(sentence, r, logprob, score))
            if exclude == k:
                break
        if not final_captions.size():
            final_captions = captions
        best = final_captions.best_first()[0]
        return best.sentence, best.logprob, best.score
