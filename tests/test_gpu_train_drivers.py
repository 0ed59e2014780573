"""The callers end to end on a tiny synthetic corpus written in the reference's own file formats: feature text
file -> C++ reader -> pinned batches -> XE training (loss falls) -> REINFORCE with the C++ CIDEr-D reward on ids
-> greedy evaluation -> checkpoint with TF variable names -> optimistic restore."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _corpus(tmp_path, name, rng, n_videos=12, d=24, tv=3):
    vocab = ["<en_unk>", "a", "man", "woman", "dog", "cat", "is", "playing", "running", "eating", "the", "guitar", "ball", "food"]
    subj, verb, obj = ["man", "woman", "dog", "cat"], ["playing", "running", "eating"], ["guitar", "ball", "food"]
    feats, sents = str(tmp_path / f"{name}_feat.txt"), str(tmp_path / f"{name}_sents.txt")
    with open(feats, "w") as f, open(sents, "w") as g:
        for v in range(n_videos):
            s, vb, o = subj[v % 4], verb[(v // 4) % 3], obj[v % 3]
            base = np.zeros(d, np.float32); base[v % 4] = 2; base[4 + (v // 4) % 3] = 2; base[8 + v % 3] = 2
            for k in range(tv):
                x = np.abs(base + 0.05 * rng.standard_normal(d)).astype(np.float32)
                f.write(f"vid{v}_frame_{k}," + ",".join(f"{t:.6f}" for t in x) + "\n")
            for cap in (f"a {s} is {vb} the {o}", f"the {s} is {vb}", f"a {s} {vb} a {o}"):
                g.write(f"vid{v}\t{cap}\n")
    return sents, feats, vocab


def test_xe_then_rl_drivers(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import s2vt_amd
    from s2vt_amd import train_common as tc, train_rl, train_xe
    rng = np.random.default_rng(0)
    sents, feats, vocab = _corpus(tmp_path, "train", rng)
    corpus = tc.Corpus(sents, feats, vocabulary=vocab)
    assert len(corpus.features) == 12 and corpus.features.features.shape == (12, 3, 24)
    cfg = tc.Config(dim_image=24, lstm_dim=32, word_dim=16, n_video_lstm_step=3, n_caption_lstm_step=8, n_epochs=6, batch_size=8,
                    start_learning_rate=2e-2, model_path=str(tmp_path / "m"), model_name="xe", step_log=str(tmp_path / "xe.jsonl"))
    quiet = lambda *_: None
    model, hist = train_xe.train(cfg, corpus, corpus, log=quiet)
    assert hist[-1]["loss"] < 0.7 * hist[0]["loss"]                      # it learns
    import json
    recs = [json.loads(l) for l in open(tmp_path / "xe.jsonl")]          # machine-readable step log: one object per step / epoch
    assert sum(r["kind"] == "epoch" for r in recs) == 6 and all("loss" in r for r in recs)
    assert [r["step"] for r in recs if r["kind"] == "step"] == list(range(1, model.global_step + 1))
    assert os.path.exists(hist[-1]["checkpoint"]) and hist[-1]["ciderD"] is not None
    rl = train_rl.rl_config(dim_image=24, lstm_dim=32, word_dim=16, n_video_lstm_step=3, n_caption_lstm_step=8, n_epochs=1,
                            batch_size=8, multisample=3, start_learning_rate=1e-3, model_path=str(tmp_path / "m"), model_name="rl")
    model2, hist2 = train_rl.train(rl, corpus, corpus, restore=hist[-1]["checkpoint"], log=quiet)
    assert np.isfinite(hist2[-1]["loss"]) and hist2[-1]["ciderD"] is not None
    # restored variables came from the XE checkpoint under the reference's TF names
    with np.load(hist[-1]["checkpoint"]) as z:
        assert "s2vt/LSTM1/basic_lstm_cell/weights" in z.files and "Wemb" in z.files


@pytest.mark.parametrize("lam", [0.0, 0.5])
def test_multitask_rl_driver(tmp_path, lam):
    """train_rl.train(attr_vocabulary=...): the multitask scripts on precomputed features (BASELINE configs[3]) -- bag-of-words labels from the
    corpus' own captions, the attribute head's term in the objective (lambda_loss = 0: reinforce_multitask_e2e_attribute_loss.py:957; 0.5: the XE mix of
    ..._attribute_s2vt.py:850), the multilabel metrics of the test loop per epoch, attr_W / attr_b in the checkpoint."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import s2vt_amd
    from s2vt_amd import hostglue, train_common as tc, train_rl
    rng = np.random.default_rng(0)
    sents, feats, vocab = _corpus(tmp_path, "train", rng)
    corpus = tc.Corpus(sents, feats, vocabulary=vocab)
    attrs = ["man", "woman", "dog", "cat", "guitar", "ball", "food", "zebra"]
    cfg = train_rl.rl_config(dim_image=24, lstm_dim=32, word_dim=16, n_video_lstm_step=3, n_caption_lstm_step=8, n_epochs=12, batch_size=8,
                             multisample=2, start_learning_rate=2e-2, model_path=str(tmp_path / "m"), model_name="mt", alpha=0.5, lambda_loss=lam)
    model, hist = train_rl.train(cfg, corpus, corpus, log=lambda *_: None, attr_vocabulary=attrs)
    assert model.label_dim == 8 and np.isfinite(hist[-1]["loss"]) and hist[-1]["ciderD"] is not None
    # the head learns the corpus' attributes (each is a function of one feature column): scores against the labels get_multilabel forms
    lab = hostglue.get_multilabel(corpus.index.by_video, attrs)
    y = np.stack([lab[v] for v in corpus.index.video_ids])
    assert y[:, 7].sum() == 0 and y[:, :4].sum(1).tolist() == [1] * 12
    sc = model.attribute_scores(corpus.features.batch(corpus.index.video_ids)).cpu().numpy()
    m = hist[-1]["multilabel"]
    assert (m["true_positive"], m["true_negative"], m["false_positive"], m["false_negative"]) == hostglue.get_metrics(sc, y, 0.5)[:4]
    assert m["true_positive"] + m["false_negative"] == int(y.sum()) and m["f1_score"] > 0.8 and m["f1_score"] > hist[0]["multilabel"].get("f1_score", 0.0)
    if lam > 0:                                                          # the XE term teaches the captions too
        assert hist[-1]["ciderD"] > 0.5
    with np.load(hist[-1]["checkpoint"]) as z:
        assert "attr_W" in z.files and "attr_b" in z.files and "attr_W/Adam" in z.files


def test_attention_driver(tmp_path):
    """train_attention.train (original_attention.py's train(), :383-520) on the synthetic corpus: the loss falls, greedy evaluation scores,
    the checkpoint carries the TF variable names and resumes."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import s2vt_amd
    from s2vt_amd import train_attention, train_common as tc
    rng = np.random.default_rng(0)
    sents, feats, vocab = _corpus(tmp_path, "att", rng)
    corpus = tc.Corpus(sents, feats, vocabulary=vocab)
    cfg = train_attention.attention_config(dim_image=24, lstm_dim=32, n_video_lstm_step=3, n_caption_lstm_step=8, n_epochs=6, batch_size=8,
                                           start_learning_rate=2e-2, model_path=str(tmp_path / "m"), model_name="att", step_log=str(tmp_path / "att.jsonl"))
    model, hist = train_attention.train(cfg, corpus, corpus, log=lambda *_: None)
    assert hist[-1]["loss"] < 0.7 * hist[0]["loss"] and hist[-1]["ciderD"] is not None
    with np.load(hist[-1]["checkpoint"]) as z:
        assert "s2vt/LSTM3/basic_lstm_cell/weights" in z.files and "embed_att_Wa" in z.files and "embed_att_Wa/Adam" in z.files and "Variable" in z.files
    cfg1 = train_attention.attention_config(dim_image=24, lstm_dim=32, n_video_lstm_step=3, n_caption_lstm_step=8, n_epochs=1, batch_size=8,
                                            start_learning_rate=2e-2, model_path=str(tmp_path / "m2"), model_name="att2")
    model2, hist2 = train_attention.train(cfg1, corpus, None, log=lambda *_: None, resume=hist[-1]["checkpoint"])
    assert model2.global_step > model.global_step - 1 and hist2[-1]["loss"] < hist[0]["loss"]


def test_e2e_driver_frames_to_checkpoint(tmp_path):
    """train_e2e.train on a synthetic frame corpus (jpg files in the reference's directory layout) with a small
    stand-in CNN: frames -> CNN -> HIP captioner -> joint update; the loss falls, both checkpoints are written."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from PIL import Image
    import s2vt_amd
    from s2vt_amd import data, train_e2e
    rng = np.random.default_rng(1)
    vocab = ["<en_unk>", "a", "red", "green", "blue", "square", "is", "shown"]
    colours = {"red": (220, 30, 30), "green": (30, 220, 30), "blue": (30, 30, 220)}
    sent = tmp_path / "sents.txt"
    with open(sent, "w") as f:
        for v in range(9):
            name = list(colours)[v % 3]
            os.makedirs(tmp_path / "frames" / f"vid{v}")
            for k in range(1, 9):
                img = np.clip(np.asarray(colours[name])[None, None, :] + rng.integers(-20, 20, (24, 24, 3)), 0, 255).astype(np.uint8)
                Image.fromarray(img).save(tmp_path / "frames" / f"vid{v}" / f"{k:06d}.jpg")
            f.write(f"vid{v}\ta {name} square is shown\nvid{v}\ta {name} square\n")
    cfg = train_e2e.e2e_config(dim_image=24, lstm_dim=32, word_dim=16, n_video_lstm_step=3, n_caption_lstm_step=8, n_epochs=8,
                               batch_size=6, start_learning_rate=2e-2, model_path=str(tmp_path / "m"))
    sents, frames = data.get_video_frame_caption_pair(str(sent), str(tmp_path / "frames"), 3)
    torch.manual_seed(0)
    cnn = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, stride=2), torch.nn.ReLU(), torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(),
                              torch.nn.Linear(8, 24), torch.nn.ReLU())
    trainer, hist = train_e2e.train(cfg, sents, frames, vocab, cnn=cnn, width=24, height=24, log=lambda *_: None)
    assert hist[-1]["loss"] < 0.8 * hist[0]["loss"]
    assert os.path.exists(hist[-1]["checkpoint"]) and os.path.exists(hist[-1]["cnn_checkpoint"])
    g = trainer.generate(torch.from_numpy(data.image_reading_processing([frames["vid0"], frames["vid1"]], 24, 24)))
    assert g.shape == (2, 8)
    # resume continues BOTH halves: a fresh, differently initialised CNN + captioner pick up the run's variables, Adam moments and
    # update counts -- the CNN's from the `-cnn-<epoch>.npz` beside the captioner's checkpoint
    import dataclasses
    torch.manual_seed(123)
    cnn2 = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, stride=2), torch.nn.ReLU(), torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(),
                               torch.nn.Linear(8, 24), torch.nn.ReLU())
    cfg0 = dataclasses.replace(cfg, n_epochs=0)
    tr2, _ = train_e2e.train(cfg0, sents, frames, vocab, cnn=cnn2, width=24, height=24, log=lambda *_: None, resume=hist[-1]["checkpoint"])
    assert train_e2e.cnn_checkpoint_of(hist[-1]["checkpoint"]) == hist[-1]["cnn_checkpoint"]
    assert torch.equal(tr2.theta, trainer.theta) and torch.equal(tr2.m, trainer.m) and torch.equal(tr2.v, trainer.v)
    assert tr2.adam_t == trainer.adam_t > 0 and float(tr2.m.abs().sum()) > 0
    assert tr2.model.global_step == trainer.model.global_step and tr2.model.adam_t == trainer.model.adam_t
    assert torch.equal(tr2.model.store.theta, trainer.model.store.theta) and torch.equal(tr2.model.store.m, trainer.model.store.m)
    assert torch.equal(tr2.generate(torch.from_numpy(data.image_reading_processing([frames["vid0"], frames["vid1"]], 24, 24))), g)
    # ... and refuses to continue with a CNN it could not restore
    with pytest.raises(FileNotFoundError, match="no CNN checkpoint"):
        train_e2e.train(cfg0, sents, frames, vocab, cnn=cnn2, width=24, height=24, log=lambda *_: None, resume=hist[-1]["checkpoint"],
                        resume_cnn=str(tmp_path / "nowhere.npz"))
    np.savez(tmp_path / "slim.npz", **{"Conv2d_1a_3x3/weights": np.zeros((3, 3, 3, 8), np.float32)})
    with pytest.raises(ValueError, match="restored 0 of"):
        train_e2e.train(cfg0, sents, frames, vocab, cnn=cnn2, width=24, height=24, log=lambda *_: None, resume=hist[-1]["checkpoint"],
                        resume_cnn=str(tmp_path / "slim.npz"))


def test_e2e_reinforce_driver(tmp_path):
    """train_e2e.train_reinforce (reinforcement_e2e.py's loop; with attr_vocabulary the multitask scripts'): frames -> one CNN forward -> sampled +
    greedy captions -> host CIDEr-D -> policy gradient through the CNN; after an XE warm start the greedy CIDEr-D holds up, the attribute head learns
    the colour words, the CNN moves, both checkpoints are written."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from PIL import Image
    import s2vt_amd
    from s2vt_amd import data, train_e2e
    rng = np.random.default_rng(1)
    vocab = ["<en_unk>", "a", "red", "green", "blue", "square", "is", "shown"]
    colours = {"red": (220, 30, 30), "green": (30, 220, 30), "blue": (30, 30, 220)}
    sent = tmp_path / "sents.txt"
    with open(sent, "w") as f:
        for v in range(9):
            name = list(colours)[v % 3]
            os.makedirs(tmp_path / "frames" / f"vid{v}")
            for k in range(1, 9):
                img = np.clip(np.asarray(colours[name])[None, None, :] + rng.integers(-20, 20, (24, 24, 3)), 0, 255).astype(np.uint8)
                Image.fromarray(img).save(tmp_path / "frames" / f"vid{v}" / f"{k:06d}.jpg")
            f.write(f"vid{v}\ta {name} square is shown\nvid{v}\ta {name} square\n")
    sents, frames = data.get_video_frame_caption_pair(str(sent), str(tmp_path / "frames"), 3)
    torch.manual_seed(0)
    cnn = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, stride=2), torch.nn.ReLU(), torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(),
                              torch.nn.Linear(8, 24), torch.nn.ReLU())
    quiet = lambda *_: None
    xe = train_e2e.e2e_config(dim_image=24, lstm_dim=32, word_dim=16, n_video_lstm_step=3, n_caption_lstm_step=8, n_epochs=10, batch_size=6,
                              start_learning_rate=2e-2, model_path=str(tmp_path / "m"))
    trainer, hist = train_e2e.train(xe, sents, frames, vocab, cnn=cnn, width=24, height=24, log=quiet)
    theta0 = trainer.theta.clone()
    rl = train_e2e.e2e_config(dim_image=24, lstm_dim=32, word_dim=16, n_video_lstm_step=3, n_caption_lstm_step=8, n_epochs=10, batch_size=6, multisample=2,
                              start_learning_rate=1e-2, model_path=str(tmp_path / "m"), model_name="e2e_rl", alpha=0.5)
    attrs = ["red", "green", "blue", "zebra"]
    trainer2, hist2 = train_e2e.train_reinforce(rl, sents, frames, vocab, cnn=cnn, width=24, height=24, log=quiet, restore=hist[-1]["checkpoint"],
                                                attr_vocabulary=attrs, test=(sents, frames))
    assert trainer2.model.label_dim == 4 and all(np.isfinite(h["loss"]) for h in hist2)
    assert hist2[-1]["ciderD"] is not None and hist2[-1]["ciderD"] > 0.3
    m = hist2[-1]["multilabel"]
    assert m["true_positive"] + m["false_negative"] == 9 and m["true_negative"] + m["false_positive"] == 27
    vids = list(dict.fromkeys(sents[:, 0].tolist()))
    sc = trainer2.evaluate_multilabel(torch.from_numpy(data.image_reading_processing([frames[v] for v in vids], 24, 24))).cpu().numpy()
    assert (sc[:, :3].argmax(1) == np.arange(9) % 3).sum() >= 7 and (sc[:, 3] < 0.5).all() and m["true_positive"] >= 6      # the head ranks the video's own colour first
    assert float((trainer2.theta - theta0).abs().max()) > 0               # the policy gradient reached the CNN
    assert os.path.exists(hist2[-1]["checkpoint"]) and os.path.exists(hist2[-1]["cnn_checkpoint"])
    with np.load(hist2[-1]["checkpoint"]) as z:
        assert "attr_W" in z.files


def test_checkpoint_resume_continues_adam_and_counters(tmp_path):
    """save -> restore into a fresh model -> one step == the uninterrupted run: the checkpoint carries Adam's moments, the
    bias-correction count, the step counter (learning-rate staircase, dropout / sampling seeds), as the reference's
    tf.train.Saver does (reinforcement_multisampling_tf_s2vt.py:661)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import s2vt_amd
    from s2vt_amd import hostglue, model as M, train_common as tc
    rng = np.random.default_rng(3)
    video = np.abs(rng.standard_normal((6, 3, 24)) * 0.5).astype(np.float32)
    cap = rng.integers(2, 40, (6, 8)).astype(np.int32); cap[:, -2:] = 0
    mask = hostglue.masks_from_ids(cap)
    cfg = tc.Config(model_path=str(tmp_path / "m"), model_name="ck", decay_steps=2, start_learning_rate=1e-2)

    def make():
        return M.Video_Caption_Generator(24, 40, 16, 32, 6, 0, 3, 8, seed=11, dropout_rate=0.9)

    def step(m):
        return m.xe_update(video, cap, mask, lr=tc.learning_rate(cfg, m.global_step), clip_norm=10.0)
    a = make()
    for _ in range(3):
        step(a)
    b = make()
    for _ in range(2):
        step(b)
    path = tc.save_checkpoint(b, cfg, 0, step_name="Variable")
    with np.load(path) as z:
        assert int(z["Variable"]) == 2 and "Wemb/Adam" in z.files and abs(float(z["beta1_power"]) - 0.9 ** 3) < 1e-6
    c = make()
    loaded = tc.optimistic_restore(c, path)
    assert c.global_step == 2 and c.adam_t == 2 and "s2vt/LSTM2/basic_lstm_cell/weights/Adam_1" in loaded
    step(c)                                                               # lr halves at step 2 (decay_steps = 2): the staircase continued
    torch.cuda.synchronize()
    ta, tc_ = a.store.theta.cpu().numpy(), c.store.theta.cpu().numpy()
    assert np.abs(ta - tc_).max() <= 5e-4                                 # (atomic-order noise of the weight-gradient reductions x Adam)
    # without the optimizer state the third step lands elsewhere: the test can tell the difference
    d = make()
    d.store.load_state_dict({k: v for k, v in b.store.state_dict().items()})
    step(d)
    assert np.abs(ta - d.store.theta.cpu().numpy()).max() > 1e-3
    # a REINFORCE run started from this XE checkpoint: the model variables only -- the reference's XE saver (tf_s2vt.py:440) holds no
    # optimizer slots, so Adam starts fresh and the staircase at 0 (:637 'g_step'); optimizer_state=True takes the slots anyway
    e = make()
    tc.optimistic_restore(e, path, step_names=("g_step",))
    assert e.global_step == 0 and e.adam_t == 0 and float(e.store.m.abs().max()) == 0.0 and torch.equal(e.store.theta, b.store.theta)
    e2 = make()
    tc.optimistic_restore(e2, path, step_names=("g_step",), optimizer_state=True)
    assert e2.global_step == 0 and e2.adam_t == 2 and torch.equal(e2.store.m, b.store.m)
    # the same through TensorFlow checkpoint FILES (V2 bundle written by save_checkpoint(checkpoint_format="tf"), tfckpt.py)
    cfg_tf = tc.Config(model_path=str(tmp_path / "mtf"), model_name="ck", decay_steps=2, start_learning_rate=1e-2, checkpoint_format="tf")
    prefix = tc.save_checkpoint(b, cfg_tf, 0, step_name="Variable")
    assert os.path.exists(prefix + ".index") and os.path.exists(prefix + ".data-00000-of-00001")
    f = make()
    tc.optimistic_restore(f, prefix)
    assert f.global_step == 2 and f.adam_t == 2
    assert torch.equal(f.store.theta, b.store.theta) and torch.equal(f.store.m, b.store.m) and torch.equal(f.store.v, b.store.v)


CHILD_DP_TRAIN = r"""
import os, sys
sys.path.insert(0, os.environ["S2VT_ROOT"])
import numpy as np, torch
sys.path.insert(0, os.path.join(os.environ["S2VT_ROOT"], "tests"))
from test_gpu_train_drivers import _corpus
import s2vt_amd
from s2vt_amd import train_common as tc, train_rl, train_xe
tmp, out, which = sys.argv[1], sys.argv[2], sys.argv[3]
rank = int(os.environ.get("RANK", "0"))
corpus = tc.Corpus(os.path.join(tmp, "train_sents.txt"), os.path.join(tmp, "train_feat.txt"), vocabulary=eval(open(os.path.join(tmp, "vocab.txt")).read()))
quiet = lambda *_: None
if which.startswith("multitask"):      # the multitask scripts' objective through train_rl (attribute labels sharded with the videos; lambda > 0: the XE mix)
    cfg = train_rl.rl_config(dim_image=24, lstm_dim=32, word_dim=16, n_video_lstm_step=3, n_caption_lstm_step=8, n_epochs=1, batch_size=8,
                             multisample=2, start_learning_rate=1e-2, max_steps_per_epoch=3, model_path=os.path.join(tmp, "m%d" % rank), model_name="mt",
                             alpha=0.3, lambda_loss=0.5 if which == "multitask_mixed" else 0.0)
    model, hist = train_rl.train(cfg, corpus, corpus, log=quiet, attr_vocabulary=["man", "woman", "dog", "cat", "guitar", "ball", "food"])
    assert "multilabel" in hist[-1] or rank != 0
elif which == "rl":
    cfg = train_rl.rl_config(dim_image=24, lstm_dim=32, word_dim=16, n_video_lstm_step=3, n_caption_lstm_step=8, n_epochs=1, batch_size=8,
                             multisample=3, start_learning_rate=1e-2, max_steps_per_epoch=3, model_path=os.path.join(tmp, "m%d" % rank), model_name="rl")
    model, hist = train_rl.train(cfg, corpus, corpus, log=quiet)
else:
    cfg = tc.Config(dim_image=24, lstm_dim=32, word_dim=16, n_video_lstm_step=3, n_caption_lstm_step=8, n_epochs=1, batch_size=8,
                    start_learning_rate=1e-2, max_steps_per_epoch=3, model_path=os.path.join(tmp, "m%d" % rank), model_name="xe")
    model, hist = train_xe.train(cfg, corpus, corpus, log=quiet)
torch.cuda.synchronize()
assert model.global_step == 3
if rank == 0:
    np.save(out, np.concatenate([model.store.theta.cpu().numpy(), [hist[-1]["ciderD"]]]))
    assert os.path.exists(hist[-1]["checkpoint"])
else:
    assert "checkpoint" not in hist[-1]
import torch.distributed as dist
if dist.is_initialized():
    dist.barrier(); dist.destroy_process_group()
print("child ok", rank)
"""


@pytest.mark.parametrize("which", ["rl", "xe", "multitask", "multitask_mixed"])
def test_data_parallel_drivers_two_ranks_equal_one(tmp_path, which):
    """train_rl.train / train_xe.train under two ranks (each B/2 of every shuffled global batch, both on this box's one GPU,
    collective over gloo -- the product's RCCL path differs only in the all_reduce call) for 3 steps end in the variables
    of one rank x B: same epoch order, global video indices in the sampling / dropout counters, sum(mask) and the clip
    after the reduce (reinforcement_multisampling_tf_s2vt.py:727-829, :643-650); rank 0 alone writes the checkpoint and
    the evaluation score is the mean over both ranks' videos."""
    import socket
    import subprocess
    import sys
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(0)
    sents, feats, vocab = _corpus(tmp_path, "train", rng, n_videos=16)
    open(tmp_path / "vocab.txt", "w").write(repr(vocab))
    outs = {}
    for world in (1, 2):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        out = str(tmp_path / f"w{world}.npy")
        base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
        base.update(S2VT_ROOT=root, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), S2VT_DIST_BACKEND="gloo")
        ps = []
        for rk in range(world):
            env = dict(base)
            if world > 1:
                env.update(RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE=str(world))
            ps.append(subprocess.Popen([sys.executable, "-c", CHILD_DP_TRAIN, str(tmp_path), out, which], env=env, stdout=subprocess.PIPE,
                                       stderr=subprocess.PIPE, text=True))
        for p in ps:
            so, se = p.communicate(timeout=600)
            assert p.returncode == 0 and "child ok" in so, f"rc={p.returncode}\n{so[-2000:]}\n{se[-4000:]}"
        outs[world] = np.load(out)
    a, b = outs[1], outs[2]
    assert np.abs(a[:-1] - b[:-1]).max() <= 5e-4, np.abs(a[:-1] - b[:-1]).max()     # 3 Adam steps at lr 1e-2: a wrong exchange moves entries by ~1e-2
    assert abs(a[-1] - b[-1]) <= 0.05 * max(1.0, abs(a[-1]))                          # CIDEr-D of the greedy captions, averaged over all videos


@pytest.mark.parametrize("mod", ["train_xe", "train_attention"])
def test_command_lines_xe_and_attention(tmp_path, mod):
    """`python -m s2vt_amd.train_xe ...` / `train_attention ...` as INTEGRATION.md gives them, at the scripts' own model dimensions."""
    import glob
    import subprocess
    import sys
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sents, feats, vocab = _corpus(tmp_path, "train", np.random.default_rng(0), n_videos=12, d=1536, tv=5)
    open(tmp_path / "vocab.txt", "w").write("\n".join(vocab) + "\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, "-m", "s2vt_amd." + mod, "--train-sents", sents, "--train-feats", feats, "--test-sents", sents, "--test-feats", feats,
           "--vocab", str(tmp_path / "vocab.txt"), "--epochs", "2", "--batch-size", "8", "--model-path", str(tmp_path / "m")]
    r = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "Epoch 1 is done" in r.stdout and glob.glob(str(tmp_path / "m" / "*"))


@pytest.mark.parametrize("launcher", ["python", "torchrun2"])
def test_command_lines_of_integration_md(tmp_path, launcher):
    """The command lines INTEGRATION.md gives: `python -m s2vt_amd.train_rl ...` (through the import alias: runpy needs the alias loader's get_code)
    and the same under `python -m torch.distributed.run --nproc-per-node 2 -m s2vt_amd.train_rl ...` (both ranks on this box's one GPU: gloo),
    at the scripts' own model dimensions, with the multitask flags: a checkpoint with the attribute head comes out, written by rank 0 alone."""
    import glob
    import socket
    import subprocess
    import sys
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(0)
    sents, feats, vocab = _corpus(tmp_path, "train", rng, n_videos=12, d=1536, tv=5)
    open(tmp_path / "vocab.txt", "w").write("\n".join(vocab) + "\n")
    open(tmp_path / "attrs.txt", "w").write("\n".join(["man", "woman", "dog", "cat", "guitar"]) + "\n")
    args = ["-m", "s2vt_amd.train_rl", "--train-sents", sents, "--train-feats", feats, "--test-sents", sents, "--test-feats", feats, "--vocab", str(tmp_path / "vocab.txt"),
            "--epochs", "1", "--batch-size", "8", "--samples", "2", "--model-path", str(tmp_path / "m"), "--attr-vocab", str(tmp_path / "attrs.txt"), "--lambda-loss", "0.5"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    if launcher == "python":
        cmd = [sys.executable] + args
    else:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        env["S2VT_DIST_BACKEND"] = "gloo"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    r = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "Epoch 0 is done" in r.stdout and "multilabel" in r.stdout and r.stdout.count("Epoch 0 is done") == 1        # rank 0 alone logs
    cks = glob.glob(str(tmp_path / "m" / "*"))
    assert cks, "no checkpoint written"


CHILD_DP_E2E_RL = r"""
import os, sys
sys.path.insert(0, os.environ["S2VT_ROOT"])
import numpy as np, torch
import s2vt_amd
from s2vt_amd import data, train_e2e
tmp, out = sys.argv[1], sys.argv[2]
rank = int(os.environ.get("RANK", "0"))
vocab = ["<en_unk>", "a", "red", "green", "blue", "square", "is", "shown"]
sents, frames = data.get_video_frame_caption_pair(os.path.join(tmp, "sents.txt"), os.path.join(tmp, "frames"), 3)
torch.manual_seed(0)
cnn = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, stride=2), torch.nn.ReLU(), torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(), torch.nn.Linear(8, 24), torch.nn.ReLU())
cfg = train_e2e.e2e_config(dim_image=24, lstm_dim=32, word_dim=16, n_video_lstm_step=3, n_caption_lstm_step=8, n_epochs=1, batch_size=8, multisample=2,
                           start_learning_rate=1e-2, max_steps_per_epoch=2, model_path=os.path.join(tmp, "m%d" % rank), alpha=0.3)
trainer, hist = train_e2e.train_reinforce(cfg, sents, frames, vocab, cnn=cnn, width=24, height=24, log=lambda *_: None,
                                          attr_vocabulary=["red", "green", "blue"], test=(sents, frames))
torch.cuda.synchronize()
assert trainer.model.global_step == 2
if rank == 0:
    m = hist[-1]["multilabel"]
    np.save(out, np.concatenate([trainer.model.store.theta.cpu().numpy(), trainer.theta.cpu().numpy(),
                                 [hist[-1]["ciderD"], m["true_positive"], m["true_negative"], m["false_positive"], m["false_negative"]]]))
    assert os.path.exists(hist[-1]["checkpoint"]) and os.path.exists(hist[-1]["cnn_checkpoint"])
else:
    assert "checkpoint" not in hist[-1]
import torch.distributed as dist
if dist.is_initialized():
    dist.barrier(); dist.destroy_process_group()
print("child ok", rank)
"""


def test_e2e_reinforce_driver_two_ranks_equal_one(tmp_path):
    """train_e2e.train_reinforce under two ranks (each B/2 videos of every global batch; gloo, one GPU) == one rank after 2 steps -- captioner AND CNN
    variables -- and the evaluation (greedy CIDEr-D, multilabel confusion counts) is summed over both ranks' videos."""
    import socket
    import subprocess
    import sys
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(1)
    colours = {"red": (220, 30, 30), "green": (30, 220, 30), "blue": (30, 30, 220)}
    with open(tmp_path / "sents.txt", "w") as f:
        for v in range(10):
            name = list(colours)[v % 3]
            os.makedirs(tmp_path / "frames" / f"vid{v}")
            for k in range(1, 9):
                img = np.clip(np.asarray(colours[name])[None, None, :] + rng.integers(-20, 20, (24, 24, 3)), 0, 255).astype(np.uint8)
                Image.fromarray(img).save(tmp_path / "frames" / f"vid{v}" / f"{k:06d}.jpg")
            f.write(f"vid{v}\ta {name} square is shown\nvid{v}\ta {name} square\n")
    outs = {}
    for world in (1, 2):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        out = str(tmp_path / f"w{world}.npy")
        base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
        base.update(S2VT_ROOT=root, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), S2VT_DIST_BACKEND="gloo")
        ps = []
        for rk in range(world):
            env = dict(base)
            if world > 1:
                env.update(RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE=str(world))
            ps.append(subprocess.Popen([sys.executable, "-c", CHILD_DP_E2E_RL, str(tmp_path), out], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        for p in ps:
            so, se = p.communicate(timeout=600)
            assert p.returncode == 0 and "child ok" in so, f"rc={p.returncode}\n{so[-2000:]}\n{se[-4000:]}"
        outs[world] = np.load(out)
    a, b = outs[1], outs[2]
    assert np.abs(a[:-5] - b[:-5]).max() <= 5e-4, np.abs(a[:-5] - b[:-5]).max()
    assert a[-4:].sum() == 30 and b[-4:].sum() == 30                    # 10 videos x 3 attributes, every one counted once
    assert abs(a[-5] - b[-5]) <= 0.05 * max(1.0, abs(a[-5]))


def test_model_state_dict_round_trip(gpu):
    """Video_Caption_Generator.state_dict() / load_state_dict() (the two methods the attention class has): TF variable names, Adam slots and both
    counters survive the round trip into a fresh model."""
    import torch
    from s2vt_amd import hostglue, model as M
    a = M.Video_Caption_Generator(24, 50, 12, 20, 4, 0, 3, 6, seed=1)
    rng = np.random.default_rng(0)
    video = np.abs(rng.standard_normal((4, 3, 24))).astype(np.float32)
    cap = rng.integers(1, 50, (4, 6)).astype(np.int32); cap[:, -1] = 0
    for _ in range(3):
        a.xe_update(video, cap, hostglue.masks_from_ids(cap), lr=1e-2)
    sd = a.state_dict()
    assert "s2vt/LSTM1/basic_lstm_cell/weights" in sd and "Wemb/Adam_1" in sd and int(sd["g_step"]) == 3
    b = M.Video_Caption_Generator(24, 50, 12, 20, 4, 0, 3, 6, seed=2)
    b.load_state_dict(sd)
    assert b.global_step == 3 and b.adam_t == 3
    assert torch.equal(a.store.theta, b.store.theta) and torch.equal(a.store.m, b.store.m) and torch.equal(a.store.v, b.store.v)
    assert set(a.state_dict(with_optimizer=False)) == {a.store.tf_names[n] for n in a.store.names}
