"""Cross-entropy training driver: the counterpart of train() in tf_s2vt.py:404-560 on the HIP path.

    python -m s2vt_amd.train_xe --train-sents S --train-feats F --test-sents S2 --test-feats F2 --vocab V

Per step the reference runs sess.run([train_op, tf_loss]) on build_model (label-smoothed XE with the Q1
batch-mean rule, weight decay, clip 10, Adam, lr 1e-3 halved every 5000 steps); here that is
Video_Caption_Generator.xe_update.  One process per GPU under torch.distributed.run for data parallel."""
from __future__ import annotations

import argparse
import random
import time

import numpy as np
import torch

from . import hostglue, reward
from .train_common import (Config, Corpus, DataParallel, StepLog, epoch_batches, greedy_eval, learning_rate, lookahead, optimistic_restore,
                           run_step, save_checkpoint, save_checkpoint_checked)


def train(cfg: Config, train_corpus: Corpus, test_corpus: Corpus | None = None, model=None, log=print, resume=None):
    """cfg.batch_size is the GLOBAL batch; data parallel as in train_rl.train (shards of each shuffled batch, Q1's per-step
    mask sums and the gradient bucket all-reduced inside xe_update, rank 0 logs and saves)."""
    from . import model as M
    par = DataParallel(model.device if model is not None else None)
    if not par.chief:
        log = lambda *_: None
    wordtoix, ixtoword = hostglue.preProBuildWordVocab(train_corpus.vocabulary)
    B = par.per_rank(cfg.batch_size)
    if model is None:
        model = M.Video_Caption_Generator(cfg.dim_image, len(wordtoix), cfg.word_dim, cfg.lstm_dim, B,
                                          cfg.n_video_lstm_step + cfg.n_caption_lstm_step, cfg.n_video_lstm_step,
                                          cfg.n_caption_lstm_step, bias_init_vector=None, seed=cfg.seed, device=par.device)
    par.attach(model)
    if resume:
        log(f"resumed: {optimistic_restore(model, resume)} at step {model.global_step}")
    scorer = reward.CiderD(test_corpus.index.refs_by_video(), wordtoix) if test_corpus is not None else None
    rng = random.Random(cfg.seed)
    caps = train_corpus.captions
    history = []
    steplog = StepLog(cfg.step_log if par.chief else None)
    def prepare(gidx):
        """Host side of one step: this rank's shard of the batch, padded captions, features staged for the copy to the GPU."""
        idx, lo = par.shard(gidx)
        vid = caps[idx, 0]
        # the GLOBAL batch is padded (64 short strings) so that every rank knows its longest caption: the steps behind
        # it are all padding on every rank and are not unrolled (Q1's batch mean couples the ranks at a live step)
        g_ind, g_mask = hostglue.sentence_padding_toix(caps[gidx, 1].tolist(), wordtoix, cfg.n_caption_lstm_step)
        g_mask = np.asarray(g_mask, np.float32)
        return dict(lo=lo, ind=np.asarray(g_ind, np.int32)[lo:lo + len(idx)], mask=g_mask[lo:lo + len(idx)],
                    steps=model.active_steps(g_mask), feats=model._dev(train_corpus.features.batch(vid), torch.float32))

    for epoch in range(cfg.n_epochs):
        losses = []
        batches = (g for it, g in enumerate(epoch_batches(len(caps), cfg.batch_size, rng)) if not (cfg.max_steps_per_epoch and it >= cfg.max_steps_per_epoch))
        cur, pending, t0 = None, None, time.time()
        for it, (gidx, gnext) in enumerate(lookahead(batches)):
            if cur is None:
                cur = prepare(gidx)
            nxt = {}

            def overlap():          # while the GPU runs this step: the next batch, and the previous step's log lines
                if gnext is not None:
                    nxt.update(prepare(gnext))
                if pending is not None:
                    pending()
            b = cur
            st, loss = run_step(model, lambda: model.xe_update(b["feats"], b["ind"], b["mask"], lr=learning_rate(cfg, model.global_step),
                                                               clip_norm=cfg.clip_norm, video_base=b["lo"], active_steps=b["steps"]),
                                log, overlap=overlap)
            losses.append(loss)
            t1 = time.time()

            def pending(it=it, loss=loss, lr=learning_rate(cfg, model.global_step), step=model.global_step, secs=t1 - t0):
                log(f"idx: {it * cfg.batch_size} rate: {lr:g} Epoch: {epoch} loss: {loss:.5f} Elapsed time: {secs:.3f}")
                steplog.write(kind="step", epoch=epoch, step=step, lr=lr, loss=loss, seconds=secs)
            t0, cur = t1, (nxt if gnext is not None else None)
        if pending is not None:
            pending()
        entry = {"epoch": epoch, "loss": float(np.mean(losses)) if losses else None}
        if test_corpus is not None:
            _, entry["ciderD"] = greedy_eval(model, test_corpus, ixtoword, scorer, B, par)
        ck = save_checkpoint_checked(model, cfg, epoch, step_name="Variable", chief=par.chief, tf_version=1)   # the XE script's saver writes V1 (tf_s2vt.py:440)      # tf_s2vt.py:441: the unnamed counter
        if par.chief:
            entry["checkpoint"] = ck
        history.append(entry)
        steplog.write(kind="epoch", **entry)
        log(f"Epoch {epoch} is done: {entry}")
    steplog.close()
    return model, history


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-sents", required=True); ap.add_argument("--train-feats", required=True)
    ap.add_argument("--test-sents"); ap.add_argument("--test-feats")
    ap.add_argument("--vocab", required=True); ap.add_argument("--resume")
    ap.add_argument("--epochs", type=int, default=30); ap.add_argument("--batch-size", type=int, default=64)
    ap.add_argument("--model-path", default="./new_s2vt_models")
    a = ap.parse_args()
    cfg = Config(n_epochs=a.epochs, batch_size=a.batch_size, model_path=a.model_path, model_name=f"batch_size{a.batch_size}_s2vt_model")
    tr = Corpus(a.train_sents, a.train_feats, vocabulary_file=a.vocab)
    te = Corpus(a.test_sents, a.test_feats, vocabulary=tr.vocabulary) if a.test_sents and a.test_feats else None
    train(cfg, tr, te, resume=a.resume)


if __name__ == "__main__":
    main()
