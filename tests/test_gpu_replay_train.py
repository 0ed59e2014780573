"""The reference's train() call sequences, replayed statement by statement through the drop-in class.

* REINFORCE: reinforcement_multisampling_tf_s2vt.py:618-652 (model + the three graphs + train_op), :743-753 (one
  sess.run on [sampled, greedy], K-1 more sampler runs), :764-806 (vstack, features x K, decode_captions_masks, baseline
  tiling), :821-826 (sess.run([train_op, sum_loss], feed_dict), sess.run(learning_rate)).
* XE: tf_s2vt.py:424-451 (model, build_model, decay + Adam + clip 10 -> train_op, build_sampler), :483-497
  (sentence_padding_toix, sess.run([train_op, tf_loss], feed_dict)).

Feeds are what the reference feeds: Python lists / numpy arrays on the host.  Every fetch is compared with the CPU
oracle (ids and logits bit-exact; losses and the post-update variables vs float64 autograd + tf.clip_by_global_norm +
TF-form Adam of oracle/s2vt_torch.py) and with the fused entry points the shim wraps.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DIMS = dict(dim_image=128, n_words=260, word_dim=32, lstm_dim=64, n_video_lstm_step=5, n_caption_lstm_step=8)
B, K = 4, 3


def _vocab():
    from s2vt_amd import hostglue
    vocabulary = ["<en_unk>"] + [f"w{i}" for i in range(DIMS["n_words"] - 3)]
    wordtoix, ixtoword = hostglue.preProBuildWordVocab(vocabulary, word_count_threshold=0)
    assert len(wordtoix) == DIMS["n_words"]
    return wordtoix, ixtoword


def _model(oracle, seed=3):
    import torch
    from s2vt_amd import model as M
    d = oracle.Dims(label_dim=0, **DIMS)
    p = oracle.init_params(d, seed=seed)
    rng = np.random.default_rng(seed + 1)
    for k in ("lstm1_b", "lstm2_b", "encode_image_b", "embed_word_b"):
        p[k] = rng.uniform(-.1, .1, p[k].shape).astype(np.float32)
    model = M.Video_Caption_Generator(dim_image=d.dim_image, n_words=d.n_words, word_dim=d.word_dim, lstm_dim=d.lstm_dim,
                                      batch_size=B, n_lstm_steps=d.n_video_lstm_step + d.n_caption_lstm_step,
                                      n_video_lstm_step=d.n_video_lstm_step, n_caption_lstm_step=d.n_caption_lstm_step,
                                      bias_init_vector=None, multisample=K)
    model.store.load(p)
    feats = np.abs(rng.standard_normal((B, d.n_video_lstm_step, d.dim_image)) * 0.5).astype(np.float32)
    features_batch = [feats[j].tolist() for j in range(B)]          # the reference feeds lists of per-frame feature lists
    return d, p, model, feats, features_batch


def _reference_update(p, grads, lr, clip):
    """tf.clip_by_global_norm + one TF-form Adam step from zero slots, float64."""
    import torch
    from oracle import s2vt_torch as T
    g, _ = T.clip_by_global_norm({k: torch.as_tensor(v) for k, v in grads.items()}, clip)
    pt = {k: torch.as_tensor(v).double() for k, v in p.items()}
    m = {k: torch.zeros_like(v) for k, v in pt.items()}
    v = {k: torch.zeros_like(x) for k, x in pt.items()}
    pt, _, _ = T.adam_tf(pt, g, m, v, 1, lr)
    return {k: x.numpy() for k, x in pt.items()}


def _check_update(model, p, ref_grads, lr, clip):
    """(1) the gradients the train_op differentiated (left in the bucket, before the clip scale) vs float64 autograd;
    (2) the variables after it vs tf.clip_by_global_norm + TF-Adam evaluated in float64 ON THOSE gradients -- Adam's first
    step is lr * g / (|g| + 3e-7), ill-conditioned in g where g ~ 0, so (2) must not inherit (1)'s tolerance."""
    g_gpu = {}
    for n in model.store.names:
        g_gpu[n] = model.store.g[n].cpu().numpy().astype(np.float64)
        scale = np.abs(ref_grads[n]).max() + 1e-30
        assert np.abs(g_gpu[n] - ref_grads[n]).max() <= 2e-4 * scale + 1e-9, n
    ref_theta = _reference_update(p, g_gpu, lr, clip)
    moved = 0.0
    for n in model.store.names:
        got = model.store.p[n].cpu().numpy().astype(np.float64)
        assert np.abs(got - ref_theta[n]).max() <= 2e-3 * lr + 1e-7, n
        moved = max(moved, np.abs(got - p[n]).max())
    assert moved > 0.5 * lr                                        # the update really happened


def test_replay_reinforce_train(gpu, oracle):
    import torch
    from s2vt_amd.hostglue import decode_captions_masks, decode_captions, masks_from_ids, tile_baseline
    from s2vt_amd.model import Session
    from oracle import s2vt_torch as T
    wordtoix, ixtoword = _vocab()
    d, p, model, feats, features_batch = _model(oracle)
    Tc, V = d.n_caption_lstm_step, d.n_words
    start_learning_rate = 1e-3

    # ---- reinforcement_multisampling_tf_s2vt.py:625-652
    _ = model.build_model()
    sampled_captions, multinomial_video_features = model.build_multinomial_sampler()
    greedy_captions, greedy_video_features = model.build_sampler()
    rewards = model.placeholder("rewards")
    base_line = model.placeholder("base_line")
    loss, loss_features, loss_captions, loss_masks = model.build_loss()
    learning_rate = model.exponential_decay(start_learning_rate, 1000, 0.5)
    train_op, sum_loss = model.reinforce_train_op((loss, loss_features, loss_captions, loss_masks), rewards, base_line,
                                                  learning_rate, clip_norm=5)
    sess = Session(model)

    # ---- :743-753
    samples, greedy_words = sess.run([sampled_captions, greedy_captions], feed_dict={
        multinomial_video_features: features_batch, greedy_video_features: features_batch})
    temps = [sess.run(sampled_captions, feed_dict={multinomial_video_features: features_batch}) for _ in range(K - 1)]
    assert samples.dtype == np.int64 and samples.shape == (B, Tc) and greedy_words.shape == (B, Tc)
    # every sampler run re-encodes and draws from its own stream: run i = oracle decode with seed_i, sample id 0
    seeds = [model.sample_seed + 7919 * (i + 1) for i in range(K)]
    for i, blk in enumerate([samples] + temps):
        ref_s, ref_g = oracle.sample_captions(p, d, feats, K=1, seed=seeds[i])
        assert np.array_equal(blk, ref_s), f"sampler run {i}"
        fused, _ = model.sample(feats, 1, False, seed=seeds[i])                  # the entry point the shim wraps
        assert np.array_equal(blk, fused.cpu().numpy())
    assert np.array_equal(greedy_words, ref_g)

    # ---- :764-806
    for t in temps:
        samples = np.vstack((samples, t))
    features_batch8 = []
    for i in range(K):
        for j in range(len(features_batch)):
            features_batch8.append(features_batch[j])
    mask, multi_decoded = decode_captions_masks(samples, ixtoword)
    greedy_mask, greedy_decoded = decode_captions_masks(greedy_words, ixtoword)
    assert multi_decoded == decode_captions(samples, ixtoword) and len(multi_decoded) == K * B
    rng = np.random.default_rng(0)
    b = rng.random(B) * 2                                        # stands in for evaluate_captions_cider (external scorer)
    b = tile_baseline(b, K)                                      # the np.vstack((b, b)) doublings of :790-795, for any K
    r = rng.random(K * B) * 2

    # the build_loss fetch itself: log p(word) * mask, [N, Tc]  (one non-zero per (n, t) of the reference's dense tensor)
    feed_dict = {loss_masks: mask, loss_captions: samples, loss_features: features_batch8, rewards: r, base_line: b}
    lp_mask = sess.run(loss, feed_dict={loss_masks: mask, loss_captions: samples, loss_features: features_batch8})
    vid = np.tile(np.arange(B, dtype=np.int32), K); sid = np.repeat(np.arange(K, dtype=np.int32), B)
    dseed = model.dropout_seed + 104729 * model.global_step
    drop = oracle.dropout_masks(dseed, vid, sid, model.dropout_rate, d.lstm_dim, d.n_video_lstm_step, Tc)
    cap32 = samples.astype(np.int32)
    ref_logits = oracle.teacher_forced(p, d, np.tile(feats, (K, 1, 1)), cap32, drop, model.dropout_rate)
    ref_lp = np.stack([oracle.row_losses(np.ascontiguousarray(ref_logits[:, t]), cap32[:, t], 0.0)[1] for t in range(Tc)], 1)
    m_arr = np.asarray(mask, np.float32)
    assert np.array_equal(m_arr, masks_from_ids(samples))
    assert lp_mask.shape == (K * B, Tc) and np.allclose(lp_mask, ref_lp * m_arr, rtol=1e-5, atol=1e-5)

    # ---- :821-826
    pt = T.to_torch(p, torch.float64, True)
    lg = T.teacher_forced(pt, torch.as_tensor(np.tile(feats, (K, 1, 1))).double(), cap32, drop, model.dropout_rate)
    ref_loss = T.pg_loss(lg, cap32, m_arr, r, b)
    ref_loss.backward()

    _, loss_val = sess.run([train_op, sum_loss], feed_dict)
    assert abs(loss_val - float(ref_loss)) <= 1e-4 * max(1.0, abs(float(ref_loss)))
    assert sess.run(learning_rate) == start_learning_rate and model.global_step == 1
    _check_update(model, p, {k: v.grad.numpy() for k, v in pt.items()}, start_learning_rate, 5.0)


def test_replay_xe_train(gpu, oracle):
    import torch
    from s2vt_amd.hostglue import sentence_padding_toix
    from s2vt_amd.model import Session
    from oracle import s2vt_torch as T
    wordtoix, ixtoword = _vocab()
    d, p, model, feats, features_batch = _model(oracle, seed=5)
    Tc = d.n_caption_lstm_step
    start_learning_rate = 1e-3

    # ---- tf_s2vt.py:432-451
    tf_loss, tf_video, tf_caption, tf_caption_mask, tf_probs = model.build_model()
    sess = Session(model)
    learning_rate = model.exponential_decay(start_learning_rate, 5000, 0.5)
    train_op = model.minimize((tf_loss, tf_video, tf_caption, tf_caption_mask, tf_probs), learning_rate, clip_norm=10)
    greedy_captions, greedy_video_features = model.build_sampler()

    # ---- :483-497
    captions_batch = ["w1 w2 w3", "w7 notaword w9 w10 w11 w12 w13 w14 w15 w16", "w5", "w200 w201 w202 w203 w204"]
    captions_ind, captions_mask = sentence_padding_toix(captions_batch, wordtoix, Tc)
    cap32 = np.asarray(captions_ind, np.int32)
    m_arr = np.asarray(captions_mask, np.float32)
    vid = np.arange(B, dtype=np.int32); sid = np.zeros(B, np.int32)
    dseed = model.dropout_seed + 104729 * model.global_step
    drop = oracle.dropout_masks(dseed, vid, sid, model.dropout_rate, d.lstm_dim, d.n_video_lstm_step, Tc)

    # forward-only fetches of the build_model graph: loss and the per-step logits
    feed_dict = {tf_video: features_batch, tf_caption: captions_ind, tf_caption_mask: captions_mask}
    loss_fwd, probs = sess.run([tf_loss, tf_probs], feed_dict)
    ref_logits = oracle.teacher_forced(p, d, feats, cap32, drop, model.dropout_rate)               # [B, Tc, V]
    assert np.array_equal(np.transpose(probs, (1, 0, 2)), ref_logits)                              # time-major list of [B, V] in the reference
    ref_fwd = oracle.xe_loss(p, d, ref_logits, cap32, m_arr, q1=True)
    assert abs(loss_fwd - ref_fwd) <= 1e-4 * max(1.0, abs(ref_fwd))

    pt = T.to_torch(p, torch.float64, True)
    lg = T.teacher_forced(pt, torch.as_tensor(feats).double(), cap32, drop, model.dropout_rate)
    ref_loss = T.xe_loss(pt, lg, cap32, m_arr, q1=True)
    ref_loss.backward()

    _, loss_val = sess.run([train_op, tf_loss], feed_dict=feed_dict)
    assert abs(loss_val - float(ref_loss)) <= 1e-4 * max(1.0, abs(float(ref_loss)))
    assert sess.run(learning_rate) == start_learning_rate and model.global_step == 1
    _check_update(model, p, {k: v.grad.numpy() for k, v in pt.items()}, start_learning_rate, 10.0)
    # the greedy graph after the update (tf_s2vt.py:508-524) still runs and equals the fused sampler on the new weights
    g = sess.run(greedy_captions, feed_dict={greedy_video_features: features_batch})
    _, fused = model.sample(feats, 0, True)
    assert np.array_equal(g, fused.cpu().numpy())
    p2 = {n: model.store.p[n].cpu().numpy() for n in model.store.names}
    assert np.array_equal(g, oracle.sample_captions(p2, d, feats, K=0, seed=0)[1])
