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


def save_cnn(trainer, cfg: Config, epoch: int):
    os.makedirs(cfg.model_path, exist_ok=True)
    path = os.path.join(cfg.model_path, f"{cfg.model_name}-cnn-{epoch}.npz")
    np.savez(path, **{k: v.detach().cpu().numpy() for k, v in trainer.cnn.state_dict().items()})
    return path


def train(cfg: Config, sents, video_frames, vocabulary, cnn=None, model=None, width=299, height=299, restore=None,
          cnn_variables=None, log=print, freeze_cnn=False, resume=None):
    """freeze_cnn: fix_e2e_tf_s2vt.py's variant (:120, :284 -- the CNN in the loop behind tf.stop_gradient; that script also runs
    batch 64 at lr 1e-3: the caller's cfg).
    restore: initialise the captioner's VARIABLES from a checkpoint of another run (an XE model, as the reference's saver.restore of
    tf_s2vt's file, which holds neither optimizer slots nor a counter, tf_s2vt.py:440): Adam starts from zero moments and the staircase
    from step 0 -- inheriting the XE run's moments and update count beside a CNN whose moments start at zero would bias-correct the
    latter as if they had seen t updates.  resume: continue THIS run -- slots, update count and step counter are taken over."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-sents", required=True); ap.add_argument("--frames", required=True)
    ap.add_argument("--vocab", required=True); ap.add_argument("--cnn-npz")
    ap.add_argument("--restore", help="variables only, from another run's checkpoint (e.g. the XE model)")
    ap.add_argument("--resume", help="a checkpoint of this run: variables, Adam slots, update count, step counter")
    ap.add_argument("--epochs", type=int, default=30); ap.add_argument("--batch-size", type=int, default=16)
    ap.add_argument("--model-path", default="./new_e2e_models")
    ap.add_argument("--freeze-cnn", action="store_true", help="fix_e2e_tf_s2vt.py: no gradient into the CNN")
    a = ap.parse_args()
    cfg = e2e_config(n_epochs=a.epochs, batch_size=a.batch_size, model_path=a.model_path)
    sents, frames = data.get_video_frame_caption_pair(a.train_sents, a.frames, cfg.n_video_lstm_step)
    variables = None
    if a.cnn_npz:
        with np.load(a.cnn_npz) as z:
            variables = {k: z[k] for k in z.files}
    train(cfg, sents, frames, data.read_vocabulary(a.vocab), restore=a.restore, resume=a.resume, cnn_variables=variables, freeze_cnn=a.freeze_cnn)


if __name__ == "__main__":
    main()
