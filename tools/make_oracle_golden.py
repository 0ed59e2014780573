#!/usr/bin/env python3
"""Freeze oracle outputs into tests/golden/oracle_small.npz (regression pin of the numeric contract:
if the oracle's arithmetic ever changes, the CPU suite fails; the GPU suite compares the HIP path to
the same frozen arrays).  Deterministic: fixed seeds, no libm calls inside the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import s2vt_oracle as orc  # noqa: E402

DIMS = dict(dim_image=40, n_words=61, word_dim=12, lstm_dim=16, n_video_lstm_step=3, n_caption_lstm_step=5, label_dim=0)


def build():
    d = orc.Dims(**DIMS)
    p = orc.init_params(d, seed=11)
    rng = np.random.default_rng(12)
    for k in ("lstm1_b", "lstm2_b", "encode_image_b", "embed_word_b"):
        p[k] = rng.uniform(-.1, .1, p[k].shape).astype(np.float32)
    B, K = 3, 2
    video = np.abs(rng.standard_normal((B, 3, 40)) * 0.5).astype(np.float32)
    s, g, slog = orc.sample_captions(p, d, video, K, seed=77, video_base=2, return_logits=True)
    N = K * B
    vid = np.tile(np.arange(B, dtype=np.int32) + 2, K); sid = np.repeat(np.arange(K, dtype=np.int32), B)
    drop = orc.dropout_masks(501, vid, sid, 0.9, d.lstm_dim, 3, 5)
    logits = orc.teacher_forced(p, d, np.tile(video, (K, 1, 1)), s, drop, 0.9)
    is_eos = s == 0
    mask = ((np.cumsum(is_eos, 1) - is_eos) == 0).astype(np.float32)
    r = rng.random(N).astype(np.float32); b = np.tile(rng.random(B).astype(np.float32), K)
    out = {"video": video, "sampled": s, "greedy": g, "sampler_logits": slog, "tf_logits": logits, "mask": mask,
           "rewards": r, "baseline": b, "pg_loss": np.float64(orc.pg_loss(logits, s, mask, r, b)),
           "xe_loss_q1": np.float64(orc.xe_loss(p, d, logits, s, mask)), "xe_loss_plain": np.float64(orc.xe_loss(p, d, logits, s, mask, q1=False))}
    out.update({"param_" + k: v for k, v in p.items()})
    return out


if __name__ == "__main__":
    path = os.path.join(ROOT, "tests", "golden", "oracle_small.npz")
    np.savez_compressed(path, **build())
    print("wrote", path, os.path.getsize(path), "bytes")
