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
