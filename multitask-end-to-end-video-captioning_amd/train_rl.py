"""Self-critical REINFORCE driver: the counterpart of train() in reinforcement_multisampling_tf_s2vt.py
:602-880 on the HIP path.

Step of the reference                                   | here
--------------------------------------------------------+---------------------------------------------------
9 sess.run of the sampler graphs (:743-753), each       | model.sample(video, K, with_greedy=True): ONE call,
  re-encoding the batch, vstack on the host (:757-764)  |   one encode, rows already sample-major
decode_captions_masks: ids -> strings + mask (:783-784) | mask on the device (s2vt_caption_mask), no strings
get_captions scan + evaluate_captions_cider x2 (:787-803| reward.CiderD.score_ids on the int32 ids (C++ inverted index, ~3 ms,
  pyciderevalcap on strings)                            |   under the GPU's teacher-forced forward); references indexed once
features tiled x8 on the host (:779-782)                | never tiled: rows address video n % B
sess.run([train_op, sum_loss]) (:823)                   | model.reinforce_update (forward with dropout, reward-scaled
                                                        |   NLL, BPTT, all-reduce, clip 5, Adam, lr 1e-6 * 0.5^(step//1000))
single GPU (:21)                                        | one process per GPU: `python -m torch.distributed.run --nproc-per-node N
                                                        |   -m s2vt_amd.train_rl ...`; batch_size stays the GLOBAL batch
"""
from __future__ import annotations

import argparse
import random
import time

import numpy as np

from . import hostglue, reward
from .train_common import (Config, Corpus, DataParallel, StepLog, epoch_batches, greedy_eval, learning_rate, lookahead, optimistic_restore,
                           run_step, save_checkpoint, save_checkpoint_checked)


def rl_config(**kw):
    base = dict(start_learning_rate=1e-6, decay_steps=1000, clip_norm=5.0, batch_size=256, multisample=8,
                model_name="reinforce_multisample_model")
    base.update(kw)
    return Config(**base)


def multilabel_eval(model, corpus, labels, batch_size, threshold=0.5):
    """The multilabel evaluation of the multitask scripts' test loop (reinforce_multitask_e2e_attribute_loss.py:1042-1073): scores of
    evaluate_multilabel on every video of `corpus` against its bag-of-words labels -> sensitivity / specificity / harmmean / precision / f1."""
    tot = np.zeros(6, np.int64)
    vids = corpus.index.video_ids
    for a in range(0, len(vids), batch_size):
        ids = vids[a:a + batch_size]
        scores = model.attribute_scores(corpus.features.batch(ids)).cpu().numpy()
        tot += np.asarray(hostglue.get_metrics(scores, labels[[corpus.index.row[v] for v in ids]], threshold), np.int64)
    tp, tn, fp, fn, cp, cn = (int(x) for x in tot)
    if tp == 0 or tn == 0:                      # (the reference's harmonic mean divides by them: an untrained head may have neither)
        return {"true_positive": tp, "true_negative": tn, "false_positive": fp, "false_negative": fn, "count_pos": cp, "count_neg": cn}
    return dict(hostglue.multilabel_summary(tp, tn, fp, fn, cp, cn), true_positive=tp, true_negative=tn, false_positive=fp, false_negative=fn)


def train(cfg: Config, train_corpus: Corpus, test_corpus: Corpus | None = None, model=None, restore=None, log=print, resume=None,
          attr_vocabulary=None):
    """attr_vocabulary (list of attribute words; None = the plain REINFORCE script): the multitask scripts on precomputed features
    (BASELINE configs[3]).  Labels = bag of words over all ground-truth captions of a video (get_multilabel, reinforce_multitask_e2e_attribute_loss.py
    :874-893), the model gets the attribute head (label_dim = len(attr_vocabulary), alpha = cfg.alpha) and the objective becomes
    -(1 - alpha) PG / sum(mask) + alpha BCE / (A B) (..._loss.py:957); with cfg.lambda_loss > 0 the ground-truth XE term of ..._s2vt.py:850 is mixed in
    (model.mixed_update: the batch's own sentences as ground truth, every variable decayed as that script's predicate does).  Each epoch also reports
    the multilabel metrics of the script's test loop (:1042-1073).
    cfg.batch_size is the GLOBAL batch (the reference's batch_size).  Under torch.distributed.run every rank walks the
    same shuffled epoch, takes its shard of each batch (videos [lo, hi) of the global batch, global indices in the noise
    counters), scores its own captions, and the gradient bucket is all-reduced inside reinforce_update; rank 0 logs and
    writes checkpoints.  restore: variables of an earlier model (optimistic, :667 -- the step counter of an XE checkpoint
    does not match this script's 'g_step', Adam's slots do); resume: a checkpoint of THIS driver, counters included."""
    import torch
    from . import model as M
    par = DataParallel(model.device if model is not None else None)
    if not par.chief:
        log = lambda *_: None
    wordtoix, ixtoword = hostglue.preProBuildWordVocab(train_corpus.vocabulary)
    K = cfg.multisample
    B = par.per_rank(cfg.batch_size)
    multitask = attr_vocabulary is not None
    if model is None:
        model = M.Video_Caption_Generator(cfg.dim_image, len(wordtoix), cfg.word_dim, cfg.lstm_dim, B,
                                          cfg.n_video_lstm_step + cfg.n_caption_lstm_step, cfg.n_video_lstm_step,
                                          cfg.n_caption_lstm_step, bias_init_vector=None, seed=cfg.seed, multisample=K, device=par.device,
                                          label_dim=len(attr_vocabulary) if multitask else 0, alpha=cfg.alpha if multitask else 0.0)
    par.attach(model)
    labels = test_labels = None
    if multitask:
        def label_matrix(corpus):
            lab = hostglue.get_multilabel(corpus.index.by_video, attr_vocabulary)
            return np.stack([lab[v] for v in corpus.index.video_ids]).astype(np.float32)
        labels = label_matrix(train_corpus)
        test_labels = label_matrix(test_corpus) if test_corpus is not None else None
    if restore:
        log(f"restored: {optimistic_restore(model, restore, step_names=('g_step',))}")
    if resume:
        log(f"resumed: {optimistic_restore(model, resume)} at step {model.global_step}")
    scorer = reward.CiderD(train_corpus.index.refs_by_video(), wordtoix)
    test_scorer = reward.CiderD(test_corpus.index.refs_by_video(), wordtoix) if test_corpus is not None else None
    rng = random.Random(cfg.seed)
    caps = train_corpus.captions
    history = []
    steplog = StepLog(cfg.step_log if par.chief else None)
    if test_corpus is not None:
        log(f"before train: ciderD {greedy_eval(model, test_corpus, ixtoword, test_scorer, B, par)[1]}")
    def prepare(gidx):
        """Host side of one step: this rank's shard of the batch, its features on their way to the GPU (one copy, shared by
        sample + update), the reward tables' rows of its videos."""
        idx, lo = par.shard(gidx)
        vid = caps[idx, 0]
        rows = np.asarray([train_corpus.index.row[v] for v in vid], np.int32)
        out = dict(lo=lo, video=model._dev(train_corpus.features.batch(vid), torch.float32), rows=rows)
        if multitask:
            out["labels"] = model._dev(labels[rows], torch.float32)
            if cfg.lambda_loss > 0:             # the batch's own sentences are the ground truth of the XE term (..._s2vt.py:977-983)
                gt, gm = hostglue.sentence_padding_toix(caps[idx, 1].tolist(), wordtoix, cfg.n_caption_lstm_step)
                out["gt"], out["gmask"] = np.asarray(gt, np.int32), np.asarray(gm, np.float32)
        return out

    for epoch in range(cfg.n_epochs):
        losses, adv = [], []
        batches = (g for it, g in enumerate(epoch_batches(len(caps), cfg.batch_size, rng)) if not (cfg.max_steps_per_epoch and it >= cfg.max_steps_per_epoch))
        cur, pending, t0 = None, None, time.time()
        for it, (gidx, gnext) in enumerate(lookahead(batches)):
            if cur is None:
                cur = prepare(gidx)
            video, rows, lo = cur["video"], cur["rows"], cur["lo"]
            rb, nxt = {}, {}

            def step():
                samples, greedy_words = model.sample(video, K, True, seed=cfg.seed + 7919 * (model.global_step + 1), video_base=lo, stop_at_eos=cfg.stop_at_eos)
                s_host, g_host = samples.cpu().numpy(), greedy_words.cpu().numpy()
                model.check_health()            # the ids just came to the host: a starved sampler recurrence is caught before it is scored

                def rewards():                  # runs on the host while the GPU does the teacher-forced forward
                    rb["r"] = scorer.score_ids(s_host, np.tile(rows, K))                 # [K*B], sample-major like the ids
                    rb["b"] = scorer.score_ids(g_host, rows)                            # [B]
                    return rb["r"], hostglue.tile_baseline(rb["b"], K)
                # the ids are on the host anyway: behind the longest sample (its first <eos> included) every position of the
                # batch is masked, and the update does not unroll those steps
                eos = s_host == 0
                steps = int(np.where(eos.any(1), eos.argmax(1) + 1, s_host.shape[1]).max())
                if multitask and cfg.lambda_loss > 0:
                    r_, b_ = rewards()
                    return model.mixed_update(video, samples, hostglue.masks_from_ids(s_host), r_, b_, cur["gt"], cur["gmask"],
                                              lr=learning_rate(cfg, model.global_step), lambda_loss=cfg.lambda_loss, clip_norm=cfg.clip_norm, video_base=lo,
                                              true_labels=cur["labels"], decay_all=True, reuse_sampler_state=True)
                if multitask:
                    return model.reinforce_update(video, samples, None, None, None, lr=learning_rate(cfg, model.global_step),
                                                  clip_norm=cfg.clip_norm, video_base=lo, reuse_sampler_state=True, reward_fn=rewards,
                                                  active_steps=steps, live_mask=hostglue.masks_from_ids(s_host), true_labels=cur["labels"])
                return model.reinforce_update(video, samples, None, None, None, lr=learning_rate(cfg, model.global_step),
                                              clip_norm=cfg.clip_norm, video_base=lo, reuse_sampler_state=True, reward_fn=rewards,
                                              active_steps=steps, live_mask=hostglue.masks_from_ids(s_host))

            def overlap():          # while the GPU runs the update: the next batch, and the previous step's log lines
                if gnext is not None:
                    nxt.update(prepare(gnext))
                if pending is not None:
                    pending()
            st, loss = run_step(model, step, log, overlap=overlap)
            r, b = rb["r"], rb["b"]
            losses.append(loss); adv.append(float(r.mean() - b.mean()))
            t1 = time.time()

            def pending(it=it, loss=loss, lr=learning_rate(cfg, model.global_step), step=model.global_step, rm=float(r.mean()), bm=float(b.mean()), secs=t1 - t0):
                log(f"idx: {it * cfg.batch_size} rate: {lr:g} Epoch: {epoch} loss: {loss:.5f} r: {rm:.4f} b: {bm:.4f} Elapsed time: {secs:.3f}")
                steplog.write(kind="step", epoch=epoch, step=step, lr=lr, loss=loss, reward=rm, baseline=bm, seconds=secs)
            t0, cur = t1, (nxt if gnext is not None else None)
        if pending is not None:
            pending()
        entry = {"epoch": epoch, "loss": float(np.mean(losses)) if losses else None, "r_minus_b": float(np.mean(adv)) if adv else None}
        if test_corpus is not None:
            _, entry["ciderD"] = greedy_eval(model, test_corpus, ixtoword, test_scorer, B, par)
            if multitask:
                entry["multilabel"] = multilabel_eval(model, test_corpus, test_labels, B)
        ck = save_checkpoint_checked(model, cfg, epoch, step_name="g_step", chief=par.chief)
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
    ap.add_argument("--vocab", required=True); ap.add_argument("--restore"); ap.add_argument("--resume")
    ap.add_argument("--epochs", type=int, default=30); ap.add_argument("--batch-size", type=int, default=256)
    ap.add_argument("--samples", type=int, default=8); ap.add_argument("--model-path", default="./new_multisamp_reinforcement_models")
    ap.add_argument("--stop-at-eos", action="store_true", help="samples leave the decode loop at their first <eos> (same update, shorter sampler loop)")
    ap.add_argument("--attr-vocab", help="multitask scripts: file with one attribute word per line (label_dim = its length)")
    ap.add_argument("--alpha", type=float, default=0.05); ap.add_argument("--lambda-loss", type=float, default=0.0)
    a = ap.parse_args()
    cfg = rl_config(n_epochs=a.epochs, batch_size=a.batch_size, multisample=a.samples, model_path=a.model_path, stop_at_eos=a.stop_at_eos,
                    alpha=a.alpha, lambda_loss=a.lambda_loss)
    tr = Corpus(a.train_sents, a.train_feats, vocabulary_file=a.vocab)
    te = Corpus(a.test_sents, a.test_feats, vocabulary=tr.vocabulary) if a.test_sents and a.test_feats else None
    attr = [l.strip() for l in open(a.attr_vocab)] if a.attr_vocab else None
    train(cfg, tr, te, restore=a.restore, resume=a.resume, attr_vocabulary=attr)


if __name__ == "__main__":
    main()
