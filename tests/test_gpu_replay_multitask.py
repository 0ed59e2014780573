"""The multitask / end-to-end scripts' train() call sequences, replayed statement by statement through the drop-in class
(the `width, height, channels, feature_dim, label_dim, alpha` constructor, the 6-tuples of build_model /
build_loss, the frame and label placeholders, the mixed train_op).

* reinforce_multitask_e2e_attribute_loss.py:922-965 (model, build_model 6-tuple with the Q5 unpacking, evaluate_multilabel,
  the two samplers, rewards / base_line, build_loss 6-tuple, exponential_decay, sum_loss = -(1-alpha) PG / norm + alpha *
  multilabel_loss, clip 10, Adam), :1085-1114 (sess.run([sampled, greedy]), decode_captions_masks, the feed_dict with the
  model_* placeholders it does not need, sess.run([train_op, sum_loss])) -- on precomputed features AND on frames with a CNN
  attached (the script's own form: frame placeholders [B, Tv, H, W, C]).
* reinforce_multitask_e2e_attribute_s2vt.py:814-858 (5- / 4-tuples, sum_loss = -(1-lambda) PG / norm + lambda * model_loss,
  clip 5), :957-978 -- features and frames.
* multitask_e2e_attribute_s2vt.py:690-717,890 (the 7-tuple build_model, clip 10 train_op,
  sess.run([train_op, tf_loss, tf_multilabel_loss])).

Every fetch is compared with the CPU oracle (ids bit-exact; losses, every gradient and the post-update variables vs float64
autograd + tf.clip_by_global_norm + TF-form Adam of oracle/s2vt_torch.py) and with the fused entry point the shim lowers
onto (a twin model: same update up to the accumulation order of the weight-gradient atomics).
"""
import copy

import numpy as np
import pytest

from test_gpu_replay_train import _check_update, _vocab

pytestmark = pytest.mark.gpu

DIMS = dict(dim_image=128, n_words=260, word_dim=32, lstm_dim=64, n_video_lstm_step=5, n_caption_lstm_step=8)
B, A = 4, 10
HW = 17


def _tiny_cnn(D, seed=0):
    import torch
    import torch.nn as nn
    torch.manual_seed(seed)
    return nn.Sequential(nn.Conv2d(3, 8, 3, stride=2), nn.ReLU(), nn.Conv2d(8, 8, 3), nn.ReLU(),
                         nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(8, D), nn.ReLU())


def _build(oracle, label_dim, alpha, seed=3, with_cnn=False, decay_all=True):
    """(dims, params, model, twin model [for the fused entry point], feature block, its list-of-lists feed, frames NHWC,
    float64 CNN copy)"""
    import torch
    from s2vt_amd import multitask
    d = oracle.Dims(label_dim=label_dim, **DIMS)
    p = oracle.init_params(d, seed=seed, attr=label_dim > 0)
    rng = np.random.default_rng(seed + 1)
    for k in ("lstm1_b", "lstm2_b", "encode_image_b", "embed_word_b"):
        p[k] = rng.uniform(-.1, .1, p[k].shape).astype(np.float32)

    def make():
        m = multitask.Video_Caption_Generator(dim_image=d.dim_image, n_words=d.n_words, word_dim=d.word_dim, lstm_dim=d.lstm_dim,
                                              batch_size=B, n_lstm_steps=d.n_video_lstm_step + d.n_caption_lstm_step,
                                              n_video_lstm_step=d.n_video_lstm_step, n_caption_lstm_step=d.n_caption_lstm_step,
                                              bias_init_vector=None, width=HW, height=HW, channels=3, feature_dim=d.dim_image,
                                              label_dim=label_dim, alpha=alpha)
        m.store.load(p)
        return m
    model, twin = make(), make()
    assert model.decay_all_variables and model.multisample == 1 and (model.width, model.height, model.channels) == (HW, HW, 3)
    feats = np.abs(rng.standard_normal((B, d.n_video_lstm_step, d.dim_image)) * 0.5).astype(np.float32)
    frames = rng.uniform(-1, 1, (B, d.n_video_lstm_step, HW, HW, 3)).astype(np.float32)        # the reference's placeholder layout
    ref_cnn = None
    if with_cnn:
        cnn = _tiny_cnn(d.dim_image, seed=seed)
        ref_cnn = copy.deepcopy(cnn).double()
        model.attach_cnn(cnn, seed=21)
        assert model.e2e is not None and model.e2e.model is model
        twin.attach_cnn(_tiny_cnn(d.dim_image, seed=seed), seed=21)
    return d, p, model, twin, feats, [feats[j].tolist() for j in range(B)], frames, ref_cnn, rng


def _ref_features(oracle, model, ref_cnn, frames, draw, keep):
    """float64 restatement of the graph's head: CNN (inference-mode batch norm: none in the stand-in) -> slim.dropout with the
    draw-th mask of the step -> [B, Tv, D]."""
    import torch
    Bn, Tv = frames.shape[:2]
    x = torch.as_tensor(frames).double().permute(0, 1, 4, 2, 3).reshape(Bn * Tv, 3, HW, HW)
    raw = ref_cnn(x)
    if draw is None or keep >= 1.0:
        return raw.reshape(Bn, Tv, -1)
    vid = np.repeat(np.arange(Bn, dtype=np.int32), Tv); frame = np.tile(np.arange(Tv, dtype=np.int32), Bn)
    seed = model.e2e.seed + 15485863 * (model.global_step + 1)
    mask = oracle.dropout_mask(seed, vid, frame, 768 + draw, keep, raw.shape[1])
    return (raw * torch.as_tensor(mask).double() / keep).reshape(Bn, Tv, -1)


def _cnn_check(model, ref_cnn, lr, clip, theta0, total_sq_ref):
    """the CNN half: gradients vs float64 autograd, ONE global norm over both halves, TF-Adam's first step on them"""
    tr = model.e2e
    ref_flat = np.concatenate([q.grad.numpy().ravel() for q in ref_cnn.parameters()])
    got = tr.grad.cpu().numpy().astype(np.float64)
    assert np.abs(got - ref_flat).max() <= 3e-4 * np.abs(ref_flat).max() + 1e-10
    assert abs(float(model._sumsq) - total_sq_ref) <= 2e-3 * total_sq_ref
    scale = min(1.0, clip / np.sqrt(float(model._sumsq)))
    g = got * scale
    m1, v1 = 0.1 * g, 0.001 * g * g
    step = lr * np.sqrt(1 - 0.999) / (1 - 0.9) * m1 / (np.sqrt(v1) + 1e-8)
    assert np.abs((theta0 - tr.theta.cpu().numpy().astype(np.float64)) - step).max() <= 2e-3 * lr + 1e-7


def _twin_equal(model, twin, lr=1e-3):
    """The train_op and the fused entry point it lowers onto ran the same launches on the same inputs: Adam's first moment (= 0.1 x the clipped
    gradient) agrees to the last bits the weight-gradient contractions' atomic accumulation order leaves open, the variables to a fraction of
    the step (lr * g / (|g| + 3e-7) is ill-conditioned where g ~ 0)."""
    def close(a, b, what):
        a, b = a.double(), b.double()
        assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-12, what
    close(model.store.m, twin.store.m, "captioner Adam m")
    assert float((model.store.theta - twin.store.theta).abs().max()) <= 2e-3 * lr
    if model.e2e is not None:
        close(model.e2e.m, twin.e2e.m, "CNN Adam m")
        assert float((model.e2e.theta - twin.e2e.theta).abs().max()) <= 2e-3 * lr
        assert model.e2e.adam_t == twin.e2e.adam_t == 1
    assert model.global_step == twin.global_step == 1 and model.adam_t == twin.adam_t == 1


@pytest.mark.parametrize("with_cnn", [False, True], ids=["features", "frames"])
def test_replay_reinforce_multitask_attribute_loss(gpu, oracle, with_cnn):
    import torch
    from s2vt_amd.hostglue import decode_captions_masks, get_metrics, masks_from_ids
    from s2vt_amd.model import Session
    from oracle import s2vt_torch as T
    wordtoix, ixtoword = _vocab()
    alpha, threshold, start_learning_rate = 0.2, 0.5, 1e-3
    d, p, model, twin, feats, features_batch, frames, ref_cnn, rng = _build(oracle, A, alpha, with_cnn=with_cnn)
    Tc, keep = d.n_caption_lstm_step, model.dropout_rate
    video_batch = frames if with_cnn else features_batch            # image_reading_processing(video_frames_batch), :1079
    clean = _ref_features(oracle, model, ref_cnn, frames, None, keep).detach().numpy().astype(np.float32) if with_cnn else feats

    # ---- reinforce_multitask_e2e_attribute_loss.py:933-965
    model_loss, model_features, model_captions, model_caption_masks, _, model_multilabel_loss = model.build_model()
    tf_test_video_frames, tf_scores = model.evaluate_multilabel(threshold=threshold)
    sampled_captions, multinomial_video_features = model.build_multinomial_sampler()
    greedy_captions, greedy_video_features = model.build_sampler()
    rewards = model.placeholder("rewards", [None])
    base_line = model.placeholder("base_line", [None])
    loss, loss_features, loss_captions, loss_masks, true_labels, multilabel_loss = model.build_loss()
    sess = Session(model)
    learning_rate = model.exponential_decay(start_learning_rate, 15000, 0.5)
    train_op, sum_loss = model.multitask_train_op((loss, loss_features, loss_captions, loss_masks, true_labels, multilabel_loss),
                                                  rewards, base_line, learning_rate, clip_norm=10, alpha=alpha)
    if with_cnn:                                                     # the placeholders ARE the frame placeholders of :118 / :232
        assert loss_features.shape == (B, d.n_video_lstm_step, HW, HW, 3) == model_features.shape and tf_test_video_frames.shape[2:] == (HW, HW, 3)
    assert true_labels.shape == (B, A) and model_multilabel_loss.shape == (B, A)      # Q5: the 6th output of build_model is ITS label placeholder

    # ---- :1031-1046 the multilabel evaluation before training
    sample_labels = (rng.random((B, A)) < 0.3).astype(np.float32).tolist()
    scores = sess.run(tf_scores, feed_dict={tf_test_video_frames: video_batch})
    ref_scores = oracle.attr_scores(p, clean)
    assert scores.shape == (B, A) and np.allclose(scores, ref_scores, rtol=1e-5, atol=1e-6)
    assert len(get_metrics(scores, sample_labels, threshold)) == 6

    # ---- :1085-1097 one multinomial sample per video + the greedy caption, then masks and rewards
    samples, greedy_words = sess.run([sampled_captions, greedy_captions], feed_dict={
        multinomial_video_features: video_batch, greedy_video_features: video_batch})
    assert samples.dtype == np.int64 and samples.shape == (B, Tc) == greedy_words.shape
    seed1 = model.sample_seed + 7919
    if with_cnn:        # slim.dropout is ON in the multinomial graph (draw 0), OFF in the greedy one
        f_s = _ref_features(oracle, model, ref_cnn, frames, 0, keep).detach().numpy().astype(np.float32)
        got_s, _ = model.e2e.extract(model._frames(frames), dropout=True, draws=(0,))
        assert np.allclose(got_s.cpu().numpy(), f_s, rtol=1e-5, atol=1e-6)
        assert np.array_equal(samples, model.sample(got_s, 1, False, seed=seed1)[0].cpu().numpy())
        assert np.array_equal(greedy_words, model.e2e.generate(model._frames(frames)).cpu().numpy())
    else:
        ref_s, ref_g = oracle.sample_captions(p, d, feats, K=1, seed=seed1)
        assert np.array_equal(samples, ref_s) and np.array_equal(greedy_words, ref_g)
    mask, multi_decoded = decode_captions_masks(samples, ixtoword)
    greedy_mask, greedy_decoded = decode_captions_masks(greedy_words, ixtoword)
    r = (rng.random(B) * 2).tolist()                                  # stand in for evaluate_captions_cider (external scorer)
    b = (rng.random(B) * 2).tolist()
    captions_ind = rng.integers(0, d.n_words, (B, Tc)).tolist()       # model_* feeds: present in the script's feed_dict, unused by sum_loss
    captions_mask = np.ones((B, Tc), np.float32)

    # the graph's own fetches: loss = log p(word) * mask, multilabel_loss = sum(bce) / (label_dim * batch)
    cap32 = samples.astype(np.int32)
    m_arr = np.asarray(mask, np.float32)
    assert np.array_equal(m_arr, masks_from_ids(samples))
    vid = np.arange(B, dtype=np.int32); sid = np.zeros(B, np.int32)
    drop = oracle.dropout_masks(model.dropout_seed + 104729 * model.global_step, vid, sid, keep, d.lstm_dim, d.n_video_lstm_step, Tc)
    pt = T.to_torch(p, torch.float64, True)
    f_l = _ref_features(oracle, model, ref_cnn, frames, 1, keep) if with_cnn else torch.as_tensor(feats).double()
    y = np.asarray(sample_labels, np.float32)
    lp_mask, ml_val = sess.run([loss, multilabel_loss], feed_dict={loss_masks: mask, loss_captions: samples, loss_features: video_batch,
                                                                   true_labels: sample_labels})
    lg = T.teacher_forced(pt, f_l, cap32, drop, keep)
    ref_lp = torch.log_softmax(lg, -1).gather(2, torch.as_tensor(cap32).long().unsqueeze(-1)).squeeze(-1).detach().numpy()
    assert lp_mask.shape == (B, Tc) and np.allclose(lp_mask, ref_lp * m_arr, rtol=1e-4, atol=1e-5)
    ref_ml = T.attr_bce(pt, f_l, y)
    assert abs(ml_val - float(ref_ml.detach())) <= 1e-5 * max(1.0, abs(float(ref_ml.detach())))

    # ---- :957 + :1113-1114
    ref_loss = (1 - alpha) * T.pg_loss(lg, cap32, m_arr, np.asarray(r, np.float32), np.asarray(b, np.float32)) + alpha * ref_ml
    ref_loss.backward()
    theta0 = model.e2e.theta.double().cpu().numpy() if with_cnn else None
    feed_dict = {loss_masks: mask, loss_captions: samples, loss_features: video_batch, rewards: r, base_line: b,
                 model_features: video_batch, model_captions: captions_ind, model_caption_masks: captions_mask, true_labels: sample_labels}
    _, loss_val = sess.run([train_op, sum_loss], feed_dict)
    assert abs(loss_val - float(ref_loss)) <= 1e-4 * max(1.0, abs(float(ref_loss)))
    assert sess.run(learning_rate) == start_learning_rate and model.global_step == 1
    grads = {k: v.grad.numpy() for k, v in pt.items()}
    assert np.abs(grads["attr_W"]).max() > 0
    if with_cnn:
        tot = sum(float((g ** 2).sum()) for g in grads.values()) + sum(float((q.grad ** 2).sum()) for q in ref_cnn.parameters())
        _cnn_check(model, ref_cnn, start_learning_rate, 10.0, theta0, tot)
        # the captioner's Adam step used the JOINT clip factor: restate it on the library's own gradients
        scale = min(1.0, 10.0 / np.sqrt(tot))
        for n in model.store.names:
            g = model.store.g[n].cpu().numpy().astype(np.float64)
            assert np.abs(g - grads[n]).max() <= 3e-4 * (np.abs(grads[n]).max() + 1e-30) + 1e-9, n
            gs = g * scale
            step = start_learning_rate * np.sqrt(1 - 0.999) / (1 - 0.9) * (0.1 * gs) / (np.sqrt(0.001 * gs * gs) + 1e-8)
            assert np.abs((p[n] - model.store.p[n].cpu().numpy().astype(np.float64)) - step).max() <= 2e-3 * start_learning_rate + 1e-7, n
    else:
        _check_update(model, p, grads, start_learning_rate, 10.0)

    # ---- the fused entry point the train_op lowers onto
    twin.alpha = alpha
    if with_cnn:
        twin.e2e.reinforce_update(torch.as_tensor(frames).permute(0, 1, 4, 2, 3), samples, mask, r, b, start_learning_rate, clip_norm=10.0,
                                  true_labels=y)
    else:
        twin.reinforce_update(feats, samples, mask, np.asarray(r, np.float32), np.asarray(b, np.float32), start_learning_rate, clip_norm=10.0,
                              true_labels=y)
    _twin_equal(model, twin)


@pytest.mark.parametrize("with_cnn", [False, True], ids=["features", "frames"])
def test_replay_reinforce_multitask_attribute_s2vt(gpu, oracle, with_cnn):
    """reinforce_multitask_e2e_attribute_s2vt.py: the attribute head is commented out there (:58-60) -- label_dim=0 -- and the
    objective mixes the cross-entropy graph in with lambda_loss."""
    import torch
    from s2vt_amd.hostglue import decode_captions_masks, sentence_padding_toix
    from s2vt_amd.model import Session
    from oracle import s2vt_torch as T
    wordtoix, ixtoword = _vocab()
    lambda_loss, start_learning_rate = 0.3, 1e-3
    d, p, model, twin, feats, features_batch, frames, ref_cnn, rng = _build(oracle, 0, 0.2, seed=7, with_cnn=with_cnn)
    Tc, keep = d.n_caption_lstm_step, model.dropout_rate
    video_batch = frames if with_cnn else features_batch

    # ---- reinforce_multitask_e2e_attribute_s2vt.py:825-858
    model_loss, model_features, model_captions, model_caption_masks, _ = model.build_model()
    sampled_captions, multinomial_video_features = model.build_multinomial_sampler()
    greedy_captions, greedy_video_features = model.build_sampler()
    rewards = model.placeholder("rewards", [None])
    base_line = model.placeholder("base_line", [None])
    loss, loss_features, loss_captions, loss_masks = model.build_loss()
    sess = Session(model)
    learning_rate = model.exponential_decay(start_learning_rate, 300000, 0.5)
    train_op, sum_loss = model.multitask_train_op((loss, loss_features, loss_captions, loss_masks), rewards, base_line, learning_rate, clip_norm=5,
                                                  build_model_outputs=(model_loss, model_features, model_captions, model_caption_masks, _),
                                                  lambda_loss=lambda_loss)

    # ---- :952-978
    captions_batch = ["w1 w2 w3", "w7 notaword w9 w10 w11 w12 w13 w14 w15 w16", "w5", "w200 w201 w202 w203 w204"]
    captions_ind, captions_mask = sentence_padding_toix(captions_batch, wordtoix, Tc)
    samples, greedy_words = sess.run([sampled_captions, greedy_captions], feed_dict={
        multinomial_video_features: video_batch, greedy_video_features: video_batch})
    mask, multi_decoded = decode_captions_masks(samples, ixtoword)
    r = (rng.random(B) * 2).tolist()
    b = (rng.random(B) * 2).tolist()

    cap32, gcap = samples.astype(np.int32), np.asarray(captions_ind, np.int32)
    m_arr, gmask = np.asarray(mask, np.float32), np.asarray(captions_mask, np.float32)
    vid = np.arange(B, dtype=np.int32)
    dseed = model.dropout_seed + 104729 * model.global_step
    drop1 = oracle.dropout_masks(dseed, vid, np.zeros(B, np.int32), keep, d.lstm_dim, d.n_video_lstm_step, Tc)
    drop2 = oracle.dropout_masks(dseed, vid, np.ones(B, np.int32), keep, d.lstm_dim, d.n_video_lstm_step, Tc)   # the ground truth = "sample" number 1
    pt = T.to_torch(p, torch.float64, True)
    if with_cnn:      # the two graphs draw independent slim.dropout masks on the pooled features (:131, :307)
        f1, f2 = _ref_features(oracle, model, ref_cnn, frames, 1, keep), _ref_features(oracle, model, ref_cnn, frames, 2, keep)
    else:
        f1 = f2 = torch.as_tensor(feats).double()
    lg1 = T.teacher_forced(pt, f1, cap32, drop1, keep)
    lg2 = T.teacher_forced(pt, f2, gcap, drop2, keep)
    model_ref = T.xe_loss(pt, lg2, gcap, gmask, q1=True, decay_all=True)
    if with_cnn:
        model_ref = model_ref + model.decay_value * sum(0.5 * (q ** 2).sum() for q in ref_cnn.parameters())
    ref_loss = (1 - lambda_loss) * T.pg_loss(lg1, cap32, m_arr, np.asarray(r, np.float32), np.asarray(b, np.float32)) + lambda_loss * model_ref
    ref_loss.backward()
    theta0 = model.e2e.theta.double().cpu().numpy() if with_cnn else None

    feed_dict = {loss_masks: mask, loss_captions: samples, loss_features: video_batch, rewards: r, base_line: b,
                 model_features: video_batch, model_captions: captions_ind, model_caption_masks: captions_mask}
    _, loss_val = sess.run([train_op, sum_loss], feed_dict)
    assert abs(loss_val - float(ref_loss)) <= 1e-4 * max(1.0, abs(float(ref_loss)))
    assert model.global_step == 1
    grads = {k: v.grad.numpy() for k, v in pt.items()}
    if with_cnn:
        tot = sum(float((g ** 2).sum()) for g in grads.values()) + sum(float((q.grad ** 2).sum()) for q in ref_cnn.parameters())
        _cnn_check(model, ref_cnn, start_learning_rate, 5.0, theta0, tot)
        for n in model.store.names:
            g = model.store.g[n].cpu().numpy().astype(np.float64)
            assert np.abs(g - grads[n]).max() <= 3e-4 * (np.abs(grads[n]).max() + 1e-30) + 1e-9, n
        twin.e2e.mixed_update(torch.as_tensor(frames).permute(0, 1, 4, 2, 3), samples, mask, r, b, captions_ind, captions_mask, start_learning_rate,
                              lambda_loss=lambda_loss, clip_norm=5.0)
    else:
        _check_update(model, p, grads, start_learning_rate, 5.0)
        twin.mixed_update(feats, samples, mask, np.asarray(r, np.float32), np.asarray(b, np.float32), captions_ind, captions_mask,
                          start_learning_rate, lambda_loss=lambda_loss, clip_norm=5.0, decay_all=True)
    _twin_equal(model, twin)


def test_mixed_update_with_its_own_ground_truth_feature_block(gpu, oracle):
    """model.mixed_update(video_gt=): the ground-truth rows read a different feature block than the sampled rows (what two
    independent slim.dropout masks give the two graphs); with video_gt == video it is the shared-block update, bit for bit."""
    import torch
    d, p, model, twin, feats, _, _, _, rng = _build(oracle, 0, 0.0, seed=9)
    Tc = d.n_caption_lstm_step
    s = rng.integers(0, d.n_words, (2 * B, Tc)).astype(np.int32); s[:, -2:] = 0
    g = rng.integers(0, d.n_words, (B, Tc)).astype(np.int32); g[:, -3:] = 0
    from s2vt_amd.hostglue import masks_from_ids
    sm, gm = masks_from_ids(s), masks_from_ids(g)
    r = rng.random(2 * B).astype(np.float32); b = np.tile(rng.random(B).astype(np.float32), 2)
    a = model.mixed_update(feats, s, sm, r, b, g, gm, 1e-3, lambda_loss=0.4, video_gt=feats.copy())
    c = twin.mixed_update(feats, s, sm, r, b, g, gm, 1e-3, lambda_loss=0.4)
    assert float(a.loss) == float(c.loss)
    _twin_equal(model, twin)
    assert model.video_grad().shape[0] == 3 * B


@pytest.mark.parametrize("with_cnn", [False, True], ids=["features", "frames"])
def test_replay_multitask_xe_train(gpu, oracle, with_cnn):
    """multitask_e2e_attribute_s2vt.py:698-717 (7-tuple build_model, clip-10 train_op over tf_loss), :880-892
    (sess.run([train_op, tf_loss, tf_multilabel_loss], feed_dict))."""
    import torch
    from s2vt_amd.hostglue import sentence_padding_toix
    from s2vt_amd.model import Session
    from oracle import s2vt_torch as T
    wordtoix, ixtoword = _vocab()
    alpha, start_learning_rate = 0.25, 1e-3
    d, p, model, twin, feats, features_batch, frames, ref_cnn, rng = _build(oracle, A, alpha, seed=5, with_cnn=with_cnn)
    Tc, keep = d.n_caption_lstm_step, model.dropout_rate
    video_batch = frames if with_cnn else features_batch

    tf_loss, tf_video, tf_caption, tf_caption_mask, tf_probs, tf_labels, tf_multilabel_loss = model.build_model(
        multilabel_normalised=True, with_multilabel_loss=True)
    tf_test_video_frames, tf_scores = model.evaluate_multilabel(threshold=0.5)
    sess = Session(model)
    learning_rate = model.exponential_decay(start_learning_rate, 60000, 0.5)
    train_op = model.minimize((tf_loss, tf_video, tf_caption, tf_caption_mask, tf_probs, tf_labels, tf_multilabel_loss), learning_rate, clip_norm=10)

    captions_batch = ["w1 w2 w3", "w7 notaword w9 w10 w11 w12 w13 w14 w15 w16", "w5", "w200 w201 w202 w203 w204"]
    captions_ind, captions_mask = sentence_padding_toix(captions_batch, wordtoix, Tc)
    sample_labels = (rng.random((B, A)) < 0.3).astype(np.float32).tolist()
    cap32, m_arr, y = np.asarray(captions_ind, np.int32), np.asarray(captions_mask, np.float32), np.asarray(sample_labels, np.float32)
    vid = np.arange(B, dtype=np.int32); sid = np.zeros(B, np.int32)
    drop = oracle.dropout_masks(model.dropout_seed + 104729 * model.global_step, vid, sid, keep, d.lstm_dim, d.n_video_lstm_step, Tc)
    pt = T.to_torch(p, torch.float64, True)
    f = _ref_features(oracle, model, ref_cnn, frames, 0, keep) if with_cnn else torch.as_tensor(feats).double()
    lg = T.teacher_forced(pt, f, cap32, drop, keep)
    xe_wd = T.xe_loss(pt, lg, cap32, m_arr, q1=True, decay_all=True)                         # XE / sum(mask) + decay * l2(every variable)
    wd = model.decay_value * sum(0.5 * (v ** 2).sum() for v in pt.values())
    if with_cnn:
        wd = wd + model.decay_value * sum(0.5 * (q ** 2).sum() for q in ref_cnn.parameters())
    ref_ml = T.attr_bce(pt, f, y, normalise=True)
    ref_loss = (1 - alpha) * (xe_wd - model.decay_value * sum(0.5 * (v ** 2).sum() for v in pt.values())) + wd + alpha * ref_ml
    ref_loss.backward()

    feed_dict = {tf_video: video_batch, tf_caption: captions_ind, tf_caption_mask: captions_mask, tf_labels: sample_labels}
    loss_fwd, ml_fwd = sess.run([tf_loss, tf_multilabel_loss], feed_dict)                    # forward-only fetches
    assert abs(loss_fwd - float(ref_loss)) <= 1e-4 * max(1.0, abs(float(ref_loss)))
    assert abs(ml_fwd - float(ref_ml)) <= 1e-5 * max(1.0, abs(float(ref_ml.detach())))
    theta0 = model.e2e.theta.double().cpu().numpy() if with_cnn else None
    _, loss_val, multilabel_loss_val = sess.run([train_op, tf_loss, tf_multilabel_loss], feed_dict=feed_dict)
    assert abs(loss_val - float(ref_loss)) <= 1e-4 * max(1.0, abs(float(ref_loss)))
    assert abs(multilabel_loss_val - float(ref_ml)) <= 1e-5 * max(1.0, abs(float(ref_ml.detach())))
    assert model.global_step == 1
    grads = {k: v.grad.numpy() for k, v in pt.items()}
    if with_cnn:
        tot = sum(float((g ** 2).sum()) for g in grads.values()) + sum(float((q.grad ** 2).sum()) for q in ref_cnn.parameters())
        _cnn_check(model, ref_cnn, start_learning_rate, 10.0, theta0, tot)
        for n in model.store.names:
            g = model.store.g[n].cpu().numpy().astype(np.float64)
            assert np.abs(g - grads[n]).max() <= 3e-4 * (np.abs(grads[n]).max() + 1e-30) + 1e-9, n
        twin.e2e.xe_step(torch.as_tensor(frames).permute(0, 1, 4, 2, 3), captions_ind, captions_mask, start_learning_rate, clip_norm=10.0,
                         true_labels=y, attr_normalised=True)
    else:
        _check_update(model, p, grads, start_learning_rate, 10.0)
        twin.xe_update(feats, captions_ind, captions_mask, start_learning_rate, clip_norm=10.0, decay_all=True, true_labels=y)
    _twin_equal(model, twin)
    # the un-normalised form of reinforce_multitask_e2e_attribute_loss.py:221-225: the 6-tuple, multilabel_loss = sum(bce)
    out6 = model.build_model()
    assert len(out6) == 6 and out6[5].shape == (B, A)
    l6 = Session(model).run(out6[0], {out6[1]: video_batch, out6[2]: captions_ind, out6[3]: captions_mask, out6[5]: sample_labels})
    assert np.isfinite(l6)


def test_frame_feed_needs_a_cnn_and_the_reference_ctor_order(gpu, oracle):
    from s2vt_amd import model as M, multitask
    from s2vt_amd.model import Session
    m = M.Video_Caption_Generator(24, 50, 8, 16, 2, 8, 3, 5, None, 1, 0.00005, 0.9, 9, 9, 3, 24, 6, 0.3)       # all eighteen, positionally
    assert (m.width, m.height, m.channels, m.feature_dim, m.label_dim, m.alpha) == (9, 9, 3, 24, 6, 0.3)
    assert "attr_W" in m.store.p and tuple(m.store.p["attr_W"].shape) == (24, 6)
    g, video = m.build_sampler()
    with pytest.raises(ValueError, match="no CNN is attached"):
        Session(m).run(g, {video: np.zeros((2, 3, 9, 9, 3), np.float32)})
    with pytest.raises(ValueError, match="feature_dim"):
        M.Video_Caption_Generator(24, 50, 8, 16, 2, 8, 3, 5, feature_dim=32, label_dim=6)
    mm = multitask.Video_Caption_Generator(1536, 50, 8, 16, 2, 8, 3, 5, device="cpu")
    assert (mm.width, mm.height, mm.channels, mm.feature_dim, mm.label_dim, mm.alpha, mm.multisample) == (299, 299, 3, 1536, 400, 0.2, 1)
    assert len(mm.build_model()) == 6 and len(mm.build_loss()) == 6 and len(mm.build_model(with_multilabel_loss=True)) == 7
