"""End-to-end training driver (CNN in the loop): the counterpart of train() in e2e_tf_s2vt.py:482-720.

    python -m s2vt_amd.train_e2e --train-sents S --frames DIR --vocab V [--cnn-npz inception_resnet_v2.npz]

Per step the reference reads B x 5 jpgs, runs sess.run([train_op, tf_loss]) on the Inception-ResNet-v2 + S2VT graph
(label-smoothed XE, weight decay on every variable, clip 10 on the joint norm, Adam, lr 1e-5 halved every 40000
steps, batch 16).  Here: data.image_reading_processing -> e2e.EndToEnd.xe_step (torch/MIOpen CNN, HIP captioner).
The slim checkpoint is read from an .npz dump of its variables (name -> array; TF itself is not in this image).
"""
from __future__ import annotations

import argparse
import os
import random
import time

import numpy as np

from . import data, hostglue
from .train_common import Config, DataParallel, epoch_batches, learning_rate, optimistic_restore, run_step, save_checkpoint, save_checkpoint_checked


def e2e_config(**kw):
    base = dict(start_learning_rate=1e-5, decay_steps=40000, clip_norm=10.0, batch_size=16, n_caption_lstm_step=35,
                model_name="e2e_s2vt_model")
    base.update(kw)
    return Config(**base)


_CNN_OPT = ("_s2vt/cnn_adam_m", "_s2vt/cnn_adam_v", "_s2vt/cnn_adam_t")


def save_cnn(trainer, cfg: Config, epoch: int):
    """The fine-tuned CNN beside the captioner's checkpoint: every tensor of the module's state_dict (weights, batch-norm beta and moving
    statistics) under its torch key, plus the CNN half of the optimizer -- the flat Adam moments and the CNN's own update count
    (e2e.EndToEnd.m / .v / .adam_t) -- so that `resume` continues BOTH halves of the joint update (the first version wrote the
    weights only and nothing read them back)."""
    os.makedirs(cfg.model_path, exist_ok=True)
    path = os.path.join(cfg.model_path, f"{cfg.model_name}-cnn-{epoch}.npz")
    sd = {k: v.detach().cpu().numpy() for k, v in trainer.cnn.state_dict().items()}
    sd[_CNN_OPT[0]] = trainer.m.detach().cpu().numpy()
    sd[_CNN_OPT[1]] = trainer.v.detach().cpu().numpy()
    sd[_CNN_OPT[2]] = np.int64(trainer.adam_t)
    np.savez(path, **sd)
    return path


def cnn_checkpoint_of(captioner_checkpoint: str):
    """The `<model_name>-cnn-<epoch>.npz` file save_cnn wrote beside the captioner checkpoint `<model_name>-<epoch>[.npz]`."""
    d, base = os.path.split(captioner_checkpoint)
    stem = base[:-4] if base.endswith(".npz") else base
    name, _, epoch = stem.rpartition("-")
    return os.path.join(d, f"{name}-cnn-{epoch}.npz")


def load_cnn(trainer, path: str):
    """Restore what save_cnn wrote into an EndToEnd trainer: module tensors by torch key (shape-checked; the parameters are views of the
    trainer's flat buffer, so the copy lands there), Adam moments and update count.  Raises when the file is missing, matches no tensor
    of this CNN, or leaves a parameter of it unset -- a resumed run must not silently pair a trained captioner with a fresh CNN."""
    import torch
    if not os.path.exists(path):
        raise FileNotFoundError(f"resume: no CNN checkpoint at {path} (save_cnn writes it beside the captioner's; pass resume_cnn= / --resume-cnn)")
    with np.load(path) as z:
        arrays = {k: z[k] for k in z.files}
    own = trainer.cnn.state_dict()
    loaded = []
    with torch.no_grad():
        for k, t in own.items():
            a = arrays.get(k)
            if a is not None and tuple(a.shape) == tuple(t.shape):
                t.copy_(torch.as_tensor(a).to(device=t.device, dtype=t.dtype))
                loaded.append(k)
    missing = [k for k, _ in trainer.cnn.named_parameters() if k not in loaded]
    if not loaded or missing:
        raise ValueError(f"resume: {path} restored {len(loaded)} of {len(own)} CNN tensors; parameters left unset: {missing[:5]}"
                         f"{' ...' if len(missing) > 5 else ''} (a TF-slim dump goes through cnn_variables= / --cnn-npz instead)")
    if all(k in arrays for k in _CNN_OPT) and arrays[_CNN_OPT[0]].shape == tuple(trainer.m.shape):
        trainer.m.copy_(torch.as_tensor(arrays[_CNN_OPT[0]]).to(trainer.m.device))
        trainer.v.copy_(torch.as_tensor(arrays[_CNN_OPT[1]]).to(trainer.v.device))
        trainer.adam_t = int(arrays[_CNN_OPT[2]])
        loaded += list(_CNN_OPT)
    else:
        raise ValueError(f"resume: {path} holds no Adam state for this CNN ({_CNN_OPT[0]} ...): it is not a checkpoint of a run to continue")
    return loaded


def train(cfg: Config, sents, video_frames, vocabulary, cnn=None, model=None, width=299, height=299, restore=None,
          cnn_variables=None, log=print, freeze_cnn=False, resume=None, resume_cnn=None):
    """freeze_cnn: fix_e2e_tf_s2vt.py's variant (:120, :284 -- the CNN in the loop behind tf.stop_gradient; that script also runs
    batch 64 at lr 1e-3: the caller's cfg).
    restore: initialise the captioner's VARIABLES from a checkpoint of another run (an XE model, as the reference's saver.restore of
    tf_s2vt's file, which holds neither optimizer slots nor a counter, tf_s2vt.py:440): Adam starts from zero moments and the staircase
    from step 0 -- inheriting the XE run's moments and update count beside a CNN whose moments start at zero would bias-correct the
    latter as if they had seen t updates.  resume: continue THIS run -- the captioner's slots, update count and step counter AND the CNN's
    weights, batch-norm statistics, Adam moments and update count (resume_cnn, default: the `-cnn-<epoch>.npz` beside `resume`; missing or
    non-matching -> an error, never a silent fresh CNN)."""
    import torch
    from . import e2e, irv2, model as M
    par = DataParallel(model.device if model is not None else None)      # cfg.batch_size is the GLOBAL batch (16 in the reference)
    if not par.chief:
        log = lambda *_: None
    wordtoix, _ = hostglue.preProBuildWordVocab(vocabulary)
    if model is None:
        model = M.Video_Caption_Generator(cfg.dim_image, len(wordtoix), cfg.word_dim, cfg.lstm_dim, par.per_rank(cfg.batch_size),
                                          cfg.n_video_lstm_step + cfg.n_caption_lstm_step, cfg.n_video_lstm_step,
                                          cfg.n_caption_lstm_step, bias_init_vector=None, seed=cfg.seed, device=par.device)
    par.attach(model)
    if restore:
        log(f"restored: {optimistic_restore(model, restore, step_names=(), optimizer_state=False)}")
    if resume:
        log(f"resumed: {optimistic_restore(model, resume, optimizer_state=True)}")
    if cnn is None:
        cnn = irv2.InceptionResnetV2()
        if cnn_variables is not None:
            log(f"cnn variables restored: {len(cnn.load_slim_checkpoint(cnn_variables))}")
    trainer = e2e.EndToEnd(model, cnn, seed=cfg.seed)
    if resume:
        log(f"cnn resumed: {len(load_cnn(trainer, resume_cnn or cnn_checkpoint_of(resume)))} tensors")
    rng = random.Random(cfg.seed)
    history = []
    for epoch in range(cfg.n_epochs):
        losses = []
        for it, gidx in enumerate(epoch_batches(len(sents), cfg.batch_size, rng)):
            if cfg.max_steps_per_epoch and it >= cfg.max_steps_per_epoch:
                break
            t0 = time.time()
            idx, lo = par.shard(gidx)
            vid, sentence = sents[idx, 0], sents[idx, 1].tolist()
            frames = torch.from_numpy(data.image_reading_processing([video_frames[v] for v in vid], width, height))
            captions_ind, captions_mask = hostglue.sentence_padding_toix(sentence, wordtoix, cfg.n_caption_lstm_step)
            st, loss = run_step(model, lambda: trainer.xe_step(frames, np.asarray(captions_ind, np.int32), captions_mask,
                                                               lr=learning_rate(cfg, model.global_step), clip_norm=cfg.clip_norm,
                                                               video_base=lo, freeze_cnn=freeze_cnn), log)
            losses.append(loss)
            log(f"idx: {it * cfg.batch_size} rate: {learning_rate(cfg, model.global_step):g} Epoch: {epoch} "
                f"loss: {losses[-1]:.5f} Elapsed time: {time.time() - t0:.3f}")
        entry = {"epoch": epoch, "loss": float(np.mean(losses)) if losses else None}
        ck = save_checkpoint_checked(model, cfg, epoch, step_name="Variable", chief=par.chief)       # (collective health check first)
        if par.chief:
            entry["checkpoint"] = ck
            entry["cnn_checkpoint"] = save_cnn(trainer, cfg, epoch)
        history.append(entry)
        log(f"Epoch {epoch} is done: {entry}")
    return trainer, history


def train_reinforce(cfg: Config, sents, video_frames, vocabulary, cnn=None, model=None, width=299, height=299, restore=None, cnn_variables=None,
                    log=print, resume=None, attr_vocabulary=None, test=None, resume_cnn=None):
    """The REINFORCE loop with the CNN in it: train() of reinforcement_e2e.py:1085-1140 (BASELINE configs[4]) and, with attr_vocabulary, of the
    multitask scripts (reinforce_multitask_e2e_attribute_loss.py:1085-1140: bag-of-words labels per video, the attribute head's term in the objective,
    the multilabel metrics of its test loop :1042-1073).  Per step: B x Tv jpgs -> ONE CNN forward -> cfg.multisample sampled + the greedy captions ->
    CIDEr-D of both against the video's own references on the host (reward.CiderD on ids) -> e2e.EndToEnd.reinforce_step (policy gradient through
    the CNN, clip cfg.clip_norm over all variables, Adam on both halves).
    test: (sents, video_frames) of the evaluation split -- greedy CIDEr-D (and the multilabel metrics) per epoch.
    restore / resume as train()."""
    import torch
    from . import e2e, irv2, model as M, reward
    par = DataParallel(model.device if model is not None else None)
    if not par.chief:
        log = lambda *_: None
    wordtoix, _ = hostglue.preProBuildWordVocab(vocabulary)
    K, B = cfg.multisample, par.per_rank(cfg.batch_size)
    multitask = attr_vocabulary is not None
    if model is None:
        model = M.Video_Caption_Generator(cfg.dim_image, len(wordtoix), cfg.word_dim, cfg.lstm_dim, B, cfg.n_video_lstm_step + cfg.n_caption_lstm_step,
                                          cfg.n_video_lstm_step, cfg.n_caption_lstm_step, bias_init_vector=None, seed=cfg.seed, multisample=K, device=par.device,
                                          label_dim=len(attr_vocabulary) if multitask else 0, alpha=cfg.alpha if multitask else 0.0)
    par.attach(model)
    if restore:
        log(f"restored: {optimistic_restore(model, restore, step_names=(), optimizer_state=False)}")
    if resume:
        log(f"resumed: {optimistic_restore(model, resume, optimizer_state=True)}")
    if cnn is None:
        cnn = irv2.InceptionResnetV2()
        if cnn_variables is not None:
            log(f"cnn variables restored: {len(cnn.load_slim_checkpoint(cnn_variables))}")
    trainer = e2e.EndToEnd(model, cnn, seed=cfg.seed)
    if resume:
        log(f"cnn resumed: {len(load_cnn(trainer, resume_cnn or cnn_checkpoint_of(resume)))} tensors")

    def side(sents_, frames_):
        index = data.CaptionIndex(sents_)
        lab = None
        if multitask:
            d = hostglue.get_multilabel(index.by_video, attr_vocabulary)
            lab = np.stack([d[v] for v in index.video_ids]).astype(np.float32)
        return index, reward.CiderD(index.refs_by_video(), wordtoix), lab, frames_
    index, scorer, labels, _ = side(sents, video_frames)
    tside = side(*test) if test is not None else None

    def evaluate():
        tindex, tscorer, tlab, tframes = tside
        vids = tindex.video_ids[par.rank::par.world] if par.world > 1 else tindex.video_ids
        total, count, tot = 0.0, 0, np.zeros(6, np.int64)
        for a in range(0, len(vids), B):
            ids = vids[a:a + B]
            rows = [tindex.row[v] for v in ids]
            frames = torch.from_numpy(data.image_reading_processing([tframes[v] for v in ids], width, height))
            g = trainer.generate(frames).cpu().numpy()
            model.check_health()
            sc = tscorer.score_ids(g, rows)
            total += float(sc.sum()); count += len(sc)
            if multitask:
                tot += np.asarray(hostglue.get_metrics(trainer.evaluate_multilabel(frames).cpu().numpy(), tlab[rows], 0.5), np.int64)
        out = {"ciderD": par.mean(total, count)}
        if multitask:
            tp, tn, fp, fn, cp, cn = (int(x) for x in par.sum_ints(tot))
            out["multilabel"] = dict(true_positive=tp, true_negative=tn, false_positive=fp, false_negative=fn)
            if tp and tn:
                out["multilabel"].update(hostglue.multilabel_summary(tp, tn, fp, fn, cp, cn))
        return out

    if tside is not None:
        log(f"before train: {evaluate()}")
    rng = random.Random(cfg.seed)
    history = []
    for epoch in range(cfg.n_epochs):
        losses, adv = [], []
        for it, gidx in enumerate(epoch_batches(len(sents), cfg.batch_size, rng)):
            if cfg.max_steps_per_epoch and it >= cfg.max_steps_per_epoch:
                break
            t0 = time.time()
            idx, lo = par.shard(gidx)
            vid = sents[idx, 0]
            rows = np.asarray([index.row[v] for v in vid], np.int32)
            frames = torch.from_numpy(data.image_reading_processing([video_frames[v] for v in vid], width, height))
            rb = {}

            def reward_fn(samples, greedy):
                rb["r"] = scorer.score_ids(samples.cpu().numpy(), np.tile(rows, K))
                rb["b"] = scorer.score_ids(greedy.cpu().numpy(), rows)
                model.check_health()
                return rb["r"], rb["b"]
            st, loss = run_step(model, lambda: trainer.reinforce_step(frames, reward_fn, lr=learning_rate(cfg, model.global_step), K=K, clip_norm=cfg.clip_norm,
                                                                      video_base=lo, true_labels=labels[rows] if multitask else None,
                                                                      sample_seed=cfg.seed + 7919 * (model.global_step + 1)), log)
            losses.append(loss); adv.append(float(rb["r"].mean() - rb["b"].mean()))
            log(f"idx: {it * cfg.batch_size} rate: {learning_rate(cfg, model.global_step):g} Epoch: {epoch} loss: {loss:.5f} "
                f"r: {rb['r'].mean():.4f} b: {rb['b'].mean():.4f} Elapsed time: {time.time() - t0:.3f}")
        entry = {"epoch": epoch, "loss": float(np.mean(losses)) if losses else None, "advantage": float(np.mean(adv)) if adv else None}
        if tside is not None:
            entry.update(evaluate())
        ck = save_checkpoint_checked(model, cfg, epoch, step_name="Variable", chief=par.chief)
        if par.chief:
            entry["checkpoint"] = ck
            entry["cnn_checkpoint"] = save_cnn(trainer, cfg, epoch)
        history.append(entry)
        log(f"Epoch {epoch} is done: {entry}")
    return trainer, history


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-sents", required=True); ap.add_argument("--frames", required=True)
    ap.add_argument("--vocab", required=True); ap.add_argument("--cnn-npz")
    ap.add_argument("--restore", help="variables only, from another run's checkpoint (e.g. the XE model)")
    ap.add_argument("--resume", help="a checkpoint of this run: variables, Adam slots, update count, step counter (+ the CNN's, from the -cnn- file beside it)")
    ap.add_argument("--resume-cnn", help="the CNN checkpoint to resume from when it is not the one beside --resume")
    ap.add_argument("--epochs", type=int, default=30); ap.add_argument("--batch-size", type=int, default=16)
    ap.add_argument("--model-path", default="./new_e2e_models")
    ap.add_argument("--freeze-cnn", action="store_true", help="fix_e2e_tf_s2vt.py: no gradient into the CNN")
    ap.add_argument("--reinforce", action="store_true", help="reinforcement_e2e.py: self-critical REINFORCE through the CNN")
    ap.add_argument("--samples", type=int, default=1); ap.add_argument("--test-sents")
    ap.add_argument("--attr-vocab", help="multitask scripts: one attribute word per line"); ap.add_argument("--alpha", type=float, default=0.05)
    a = ap.parse_args()
    cfg = e2e_config(n_epochs=a.epochs, batch_size=a.batch_size, model_path=a.model_path, multisample=a.samples, alpha=a.alpha,
                     **({"model_name": "e2e_reinforce_model"} if a.reinforce else {}))       # (its own files beside the XE run's in one --model-path)
    sents, frames = data.get_video_frame_caption_pair(a.train_sents, a.frames, cfg.n_video_lstm_step)
    variables = None
    if a.cnn_npz:
        with np.load(a.cnn_npz) as z:
            variables = {k: z[k] for k in z.files}
    if a.reinforce:
        test = data.get_video_frame_caption_pair(a.test_sents, a.frames, cfg.n_video_lstm_step) if a.test_sents else None
        attr = [l.strip() for l in open(a.attr_vocab)] if a.attr_vocab else None
        train_reinforce(cfg, sents, frames, data.read_vocabulary(a.vocab), restore=a.restore, resume=a.resume, cnn_variables=variables,
                        attr_vocabulary=attr, test=test, resume_cnn=a.resume_cnn)
        return
    train(cfg, sents, frames, data.read_vocabulary(a.vocab), restore=a.restore, resume=a.resume, cnn_variables=variables, freeze_cnn=a.freeze_cnn,
          resume_cnn=a.resume_cnn)


if __name__ == "__main__":
    main()
