"""Pieces shared by the two training drivers (train_xe.py / train_rl.py): configuration, corpus loading,
the epoch iterator of the reference's train() loops, greedy evaluation, checkpoints."""
from __future__ import annotations

import os
import random
from dataclasses import dataclass

import numpy as np

from . import data, hostglue, reward


@dataclass
class Config:
    # the "Train Parameters" blocks of tf_s2vt.py:304-321 / reinforcement_multisampling_tf_s2vt.py:505-517
    dim_image: int = 1536
    lstm_dim: int = 1000
    word_dim: int = 500
    n_video_lstm_step: int = 5
    n_caption_lstm_step: int = 35
    n_epochs: int = 30
    batch_size: int = 64
    start_learning_rate: float = 1e-3
    decay_steps: int = 5000            # tf.train.exponential_decay(lr, step, decay_steps, 0.5, staircase=True)
    clip_norm: float = 10.0
    multisample: int = 8               # RL: sampler passes per step (K)
    seed: int = 4
    model_path: str = "./models"
    model_name: str = "s2vt_model"
    max_steps_per_epoch: int = 0       # 0 = the whole epoch (tests bound it)
    step_log: str = ""                 # path of a JSONL step log ("" = none)


class Corpus:
    def __init__(self, sent_file, feature_file, vocabulary=None, vocabulary_file=None):
        self.captions, self.features = data.get_video_feature_caption_pair(sent_file, feature_file)
        if vocabulary is None and vocabulary_file is not None:
            vocabulary = data.read_vocabulary(vocabulary_file)
        self.vocabulary = vocabulary
        self.index = data.CaptionIndex(self.captions)


class StepLog:
    """One JSON object per line per training step / epoch (the reference prints and appends to loss.txt, tf_s2vt.py:463-499):
    machine-readable, flushed per record, safe to tail.  path=None disables it."""

    def __init__(self, path=None):
        self._f = open(path, "a") if path else None

    def write(self, **record):
        if self._f:
            import json
            self._f.write(json.dumps(record) + "\n")
            self._f.flush()

    def close(self):
        if self._f:
            self._f.close()
            self._f = None


def learning_rate(cfg: Config, global_step: int) -> float:
    return cfg.start_learning_rate * 0.5 ** (global_step // cfg.decay_steps)


def epoch_batches(n_items: int, batch_size: int, rng: random.Random):
    """The reference's iteration: shuffle all (video, sentence) pairs, walk full batches only
    (zip(range(0, n - B, B), range(B, n, B)), tf_s2vt.py:477-482)."""
    index = list(range(n_items))
    rng.shuffle(index)
    for start, end in zip(range(0, n_items - batch_size, batch_size), range(batch_size, n_items, batch_size)):
        yield index[start:end]


def greedy_eval(model, corpus: Corpus, ixtoword, scorer: "reward.CiderD | None", batch_size: int):
    """Greedy captions for every test video (tf_s2vt.py:508-524) and their mean CIDEr-D against the video's own
    references when a scorer over that corpus is given (the reference reports BLEU/METEOR/ROUGE/CIDEr through
    the external coco-caption package, which is not part of this build)."""
    vids = corpus.index.video_ids
    decoded, scores = {}, []
    for a in range(0, len(vids), batch_size):
        ids = vids[a:a + batch_size]
        _, g = model.sample(corpus.features.batch(ids), 0, True)
        g = g.cpu().numpy()
        for v, s in zip(ids, hostglue.decode_captions(g, ixtoword)):
            decoded[v] = s
        if scorer is not None:
            scores.append(scorer.score_ids(g, [corpus.index.row[v] for v in ids]))
    return decoded, (float(np.concatenate(scores).mean()) if scores else None)


def save_checkpoint(model, cfg: Config, epoch: int):
    os.makedirs(cfg.model_path, exist_ok=True)
    path = os.path.join(cfg.model_path, f"{cfg.model_name}-{epoch}.npz")
    np.savez(path, **model.store.state_dict())           # keys = the reference's TF variable names
    return path


def optimistic_restore(model, path):
    """Load every variable whose name and shape match (reinforcement_multisampling_tf_s2vt.py:47-61)."""
    with np.load(path) as z:
        return model.store.load_state_dict({k: z[k] for k in z.files})
