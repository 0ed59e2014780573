"""Training driver of the temporal-attention captioner: the counterpart of train() in original_attention.py:383-520 on the HIP path.

    python -m s2vt_amd.train_attention --train-sents S --train-feats F --test-sents S2 --test-feats F2 --vocab V [--frames 32]

Per step the reference runs sess.run([train_op, tf_loss]) on build_model (cross entropy on the ground-truth caption + the alpha
regulariser, clip 10, Adam, lr 1e-4 halved every 10000 steps, :430-441); here that is Attention_Caption_Generator.xe_update.  Every
epoch: greedy captions of the test videos through the sampler graph (:483-497) and a checkpoint under the TF variable names (:519).
One process per GPU under torch.distributed.run for data parallel (every rank walks the same shuffled epoch and takes its shard)."""
from __future__ import annotations

import argparse
import random
import time

import numpy as np
import torch

from . import hostglue, reward
from .train_common import (Config, Corpus, DataParallel, StepLog, epoch_batches, greedy_eval, learning_rate, lookahead, optimistic_restore,
                           run_step, save_checkpoint_checked)


def attention_config(**kw):
    """original_attention.py:293-311: dim_hidden 1000, 35 caption steps, 20 epochs, lr 1e-4 / 10000 steps, clip 10."""
    base = dict(start_learning_rate=1e-4, decay_steps=10000, clip_norm=10.0, batch_size=64, n_epochs=20, n_caption_lstm_step=35,
                model_path="./attention_models", model_name="attention_model")
    base.update(kw)
    return Config(**base)


def train(cfg: Config, train_corpus: Corpus, test_corpus: Corpus | None = None, model=None, log=print, resume=None, m=0.5, beta=10.0):
    """cfg.batch_size is the GLOBAL batch; cfg.lstm_dim is dim_hidden (the word embedding has the same width, :65)."""
    from . import attention as A
    par = DataParallel(model.device if model is not None else None)
    if not par.chief:
        log = lambda *_: None
    wordtoix, ixtoword = hostglue.preProBuildWordVocab(train_corpus.vocabulary)
    B = par.per_rank(cfg.batch_size)
    if model is None:
        model = A.Attention_Caption_Generator(cfg.dim_image, len(wordtoix), cfg.lstm_dim, B, cfg.n_video_lstm_step, cfg.n_caption_lstm_step, 0.9,
                                              bias_init_vector=None, m=m, beta=beta, seed=cfg.seed, device=par.device)
    par.attach(model)
    if resume:
        log(f"resumed: {optimistic_restore(model, resume)} at step {model.global_step}")
    scorer = reward.CiderD(test_corpus.index.refs_by_video(), wordtoix) if test_corpus is not None else None
    rng = random.Random(cfg.seed)
    caps = train_corpus.captions
    history = []
    steplog = StepLog(cfg.step_log if par.chief else None)

    def prepare(gidx):
        idx, lo = par.shard(gidx)
        vid = caps[idx, 0]
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
        ck = save_checkpoint_checked(model, cfg, epoch, step_name="Variable", chief=par.chief)
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
    ap.add_argument("--epochs", type=int, default=20); ap.add_argument("--batch-size", type=int, default=64)
    ap.add_argument("--frames", type=int, default=5, help="n_video_lstm_steps (the feature file's frames per video; up to 64)")
    ap.add_argument("--model-path", default="./attention_models")
    a = ap.parse_args()
    cfg = attention_config(n_epochs=a.epochs, batch_size=a.batch_size, model_path=a.model_path, n_video_lstm_step=a.frames,
                           model_name=f"batch_size{a.batch_size}_beta10_m05_{a.frames}img_attention_model")
    tr = Corpus(a.train_sents, a.train_feats, vocabulary_file=a.vocab)
    te = Corpus(a.test_sents, a.test_feats, vocabulary=tr.vocabulary) if a.test_sents and a.test_feats else None
    train(cfg, tr, te, resume=a.resume)


if __name__ == "__main__":
    main()
