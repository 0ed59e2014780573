"""GPU parity of the temporal-attention captioner as a whole model (original_attention.py:53-251), through the C ABI
(s2vt_attn_teacher_forced_fwd / s2vt_attn_bptt_bwd / s2vt_attn_decode_greedy / s2vt_attn_step_scalars):

* logits, alphas and greedy ids bit-exact vs the CPU oracle (oracle/s2vt_oracle.py::attention_forward), with and
  without the DropoutWrapper masks, at Tv = 5 (the script's default), Tv > 8 (the alpha regulariser is live), Tv > 32 (the
  score kernel's frame chunking) and at the full dimensions B = 64, H = 1000, |V| = 12000, Tc = 20;
* loss and every gradient vs float64 autograd of oracle/s2vt_torch.py::attention_xe_loss (regulariser included);
* the reference's train() statements (:403-441, :463-470) replayed through build_model / build_sampler / build_generator +
  Session.run.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(oracle, D, V, H, Tv, Tc, B, seed, model_kw=None):
    from s2vt_amd import attention as A
    d = oracle.Dims(dim_image=D, n_words=V, word_dim=0, lstm_dim=H, n_video_lstm_step=Tv, n_caption_lstm_step=Tc, label_dim=0)
    p = oracle.init_attention_params(d, seed)
    rng = np.random.default_rng(seed + 1)
    for k in ("lstm3_b", "embed_att_ba", "embed_nn_bp", "embed_word_b", "encode_image_b"):
        p[k] = rng.uniform(-.1, .1, p[k].shape).astype(np.float32)
    m = A.Attention_Caption_Generator(D, V, H, B, Tv, Tc, 0.9, **(model_kw or {}))
    m.load(p)
    video = np.abs(rng.standard_normal((B, Tv, D)) * 0.5).astype(np.float32)
    cap = rng.integers(0, V, (B, Tc)).astype(np.int32)
    return d, p, m, video, cap, rng


def _drop(oracle, seed, B, H, Tc, keep):
    vid = np.arange(B, dtype=np.int32); sid = np.zeros(B, np.int32)
    return [oracle.dropout_mask(seed, vid, sid, 768 + t, keep, H) for t in range(Tc)]


@pytest.mark.parametrize("D,V,H,Tv,Tc,B,keep", [(48, 131, 32, 5, 6, 5, 1.0), (48, 131, 32, 5, 6, 5, 0.9), (40, 97, 36, 12, 4, 7, 0.9),
                                                (24, 61, 20, 40, 3, 3, 1.0), (30, 77, 30, 9, 5, 4, 0.5)])
def test_attention_model_forward_and_greedy_bit_exact(gpu, oracle, D, V, H, Tv, Tc, B, keep):
    d, p, m, video, cap, _ = _setup(oracle, D, V, H, Tv, Tc, B, 3)
    seed = 321
    drop = None if keep >= 1.0 else _drop(oracle, seed, B, H, Tc, keep)
    ref_l, ref_a, _ = oracle.attention_forward(p, d, video, cap, drop, keep)
    lg, al, _ = m.forward(video, cap, keep=keep, seed=seed)
    assert np.array_equal(al.cpu().numpy(), ref_a)
    assert np.array_equal(lg.cpu().numpy(), ref_l)
    _, ref_ga, ref_ids = oracle.attention_forward(p, d, video, None, greedy=True)
    _, ga, ids = m.forward(video, None, greedy=True)
    assert np.array_equal(ids.cpu().numpy(), ref_ids) and np.array_equal(ga.cpu().numpy(), ref_ga)


def _ref_loss_grads(oracle, p, video, cap, mask, drop, keep, beta, mm):
    import torch
    from oracle import s2vt_torch as T
    pt = T.to_torch(p, torch.float64, True)
    loss, _, alphas = T.attention_xe_loss(pt, torch.as_tensor(video).double(), cap, mask, drop, keep, beta, mm)
    loss.backward()
    return float(loss), {k: v.grad.numpy() for k, v in pt.items()}, alphas.detach().numpy()


def _check_grads(m, ref_g, tol=2e-4):
    for k in m.store.names:
        ref = ref_g[k].reshape(m.store.shapes[k])
        got = m.store.g[k].cpu().numpy().astype(np.float64)
        assert np.abs(got - ref).max() <= tol * (np.abs(ref).max() + 1e-12) + 1e-9, k


@pytest.mark.parametrize("Tv,steps", [(5, None), (32, None), (12, 4)])
def test_attention_training_gradients_vs_float64_autograd(gpu, oracle, Tv, steps):
    """loss (regulariser included) and d(loss)/d(every variable) vs float64 autograd with the same Philox dropout masks (the
    dropped LSTM3 output also feeds the next step's attention query, :135).  Tv = 5: the hinge is closed (sum(alpha) = 1 > m);
    Tv = 32 / 12: the first 8 frames hold less than m = 0.5 of the mass on most rows, the regulariser and its gradient are live.
    steps: a truncated unroll (every position behind it masked)."""
    D, V, H, Tc, B, keep = 48, 131, 32, 6, 5, 0.9
    d, p, m, video, cap, rng = _setup(oracle, D, V, H, Tv, Tc, B, 4, dict(m=0.5 if Tv != 12 else 0.8))
    mask = (rng.random((B, Tc)) < 0.8).astype(np.float32); mask[:, 0] = 1
    if steps:
        mask[:, steps:] = 0
    seed = m.dropout_seed
    ref_loss, ref_g, alphas = _ref_loss_grads(oracle, p, video, cap, mask, _drop(oracle, seed, B, H, Tc, keep), keep, m.beta, m.m)
    if Tv > 8:
        assert (alphas[:, :8, :].sum(1) < m.m).any(), "the test must exercise the open hinge"
    st = m.xe_update(video, cap, mask, lr=0.0, keep=keep)
    assert abs(float(st.loss) - ref_loss) < 1e-5 * max(1.0, abs(ref_loss))
    _check_grads(m, ref_g)
    assert float(m.loss(video, cap, mask, keep=keep)) != 0.0
    # and it trains: a few Adam steps lower the loss on the same batch
    l0 = float(m.xe_update(video, cap, mask, lr=1e-2, keep=1.0).loss)
    for _ in range(8):
        l1 = float(m.xe_update(video, cap, mask, lr=1e-2, keep=1.0).loss)
    assert l1 < l0


def test_attention_regulariser_value_vs_oracle(gpu, oracle):
    """beta * max(0, m - sum(alpha[:, 0:8])) * mask (:123,144): the forward's first-8-frames sums and the loss scalar vs the oracle's
    restatement on the oracle's own (bit-identical) alphas."""
    d, p, m, video, cap, rng = _setup(oracle, 40, 97, 36, 20, 5, 6, 9)
    mask = (rng.random((6, 5)) < 0.7).astype(np.float32); mask[:, 0] = 1
    ref_l, ref_a, _ = oracle.attention_forward(p, d, video, cap)
    assert oracle.attention_regulariser(ref_a, mask, m.beta, m.m).sum() > 0
    ref = oracle.attention_xe_loss(ref_l, ref_a, cap, mask, m.beta, m.m)
    got = float(m.loss(video, cap, mask, keep=1.0))
    assert abs(got - ref) <= 1e-5 * abs(ref)


FULL = dict(D=1536, V=12000, H=1000, Tc=20, B=64)


@pytest.mark.parametrize("Tv", [5, 32])
def test_attention_fullsize_bit_exact_and_gradients(gpu, oracle, Tv):
    """B = 64, H = 1000, |V| = 12000, Tc = 20 (the bench shape; Tv = 32 = the '32img' model of original_attention.py:287-290, at
    B = 16 / Tc = 8 to bound the oracle's time): logits, alphas and greedy ids bit-exact vs the oracle; loss and every gradient
    of one training step vs float64 autograd."""
    import torch
    f = dict(FULL)
    if Tv == 32:
        f.update(B=16, Tc=8, V=4000)
    d, p, m, video, cap, rng = _setup(oracle, f["D"], f["V"], f["H"], Tv, f["Tc"], f["B"], 21)
    B, Tc, H = f["B"], f["Tc"], f["H"]
    ln = 1 + np.minimum(rng.poisson(6, B), Tc - 2)
    for i in range(B):
        cap[i, ln[i]:] = 0
    from s2vt_amd import hostglue
    mask = hostglue.masks_from_ids(cap)
    keep, seed = 0.9, m.dropout_seed
    drop = _drop(oracle, seed, B, H, Tc, keep)
    ref_l, ref_a, _ = oracle.attention_forward(p, d, video, cap, drop, keep)
    lg, al, _ = m.forward(video, cap, keep=keep, seed=seed)
    assert np.array_equal(al.cpu().numpy(), ref_a)
    assert np.array_equal(lg.cpu().numpy(), ref_l)
    _, ref_ga, ref_ids = oracle.attention_forward(p, d, video, None, greedy=True)
    _, ga, ids = m.forward(video, None, greedy=True)
    assert np.array_equal(ids.cpu().numpy(), ref_ids) and np.array_equal(ga.cpu().numpy(), ref_ga)
    ref_loss, ref_g, _ = _ref_loss_grads(oracle, p, video, cap, mask, drop, keep, m.beta, m.m)
    st = m.xe_update(video, cap, mask, lr=0.0, keep=keep, active_steps=None)
    assert abs(float(st.loss) - ref_loss) <= 1e-3 * max(1.0, abs(ref_loss))
    _check_grads(m, ref_g)
    gn = sum(float((g ** 2).sum()) for g in ref_g.values())
    assert abs(float(st.grad_sumsq) - gn) <= 1e-3 * gn
    # the truncated unroll ("auto": the host mask's longest caption) gives the same update
    g_full = {k: m.store.g[k].clone() for k in m.store.names}
    m.global_step = 0
    st2 = m.xe_update(video, cap, mask, lr=0.0, keep=keep, active_steps="auto")
    assert abs(float(st2.loss) - float(st.loss)) <= 1e-6 * abs(float(st.loss))
    for k in m.store.names:
        a, b = m.store.g[k], g_full[k]
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-12, k


def _hold(gpu, on):
    import contextlib
    return gpu.chain_hold() if on else contextlib.nullcontext()


@pytest.mark.parametrize("hold", [False, True])
def test_attention32_benched_shape_forward_bit_exact(gpu, oracle, hold):
    """The shape `bench.py --workload attention32` times and profiles -- the '32img' model of original_attention.py:287-290 at B = 64,
    Tv = 32, H = 1000, Tc = 20, |V| = 12000: frame chunking at 20 frames per chunk, 64 attention workgroups -- logits, alphas and greedy
    ids array_equal vs the C oracle, in the persistent form (attn_chain.hip) and in the per-step launches (`chain_hold`)."""
    f = dict(FULL)
    Tv = 32
    d, p, m, video, cap, rng = _setup(oracle, f["D"], f["V"], f["H"], Tv, f["Tc"], f["B"], 33)
    B, Tc, H = f["B"], f["Tc"], f["H"]
    ln = 1 + np.minimum(rng.poisson(6, B), Tc - 2)
    for i in range(B):
        cap[i, ln[i]:] = 0
    keep, seed = 0.9, m.dropout_seed
    drop = _drop(oracle, seed, B, H, Tc, keep)
    ref_l, ref_a, _ = oracle.attention_forward(p, d, video, cap, drop, keep)
    assert ref_a.shape[1] == Tv and (ref_a[:, :8, :].sum(1) < m.m).any()        # the regulariser's hinge is open on this shape
    gpu.prof_filter(-1, -1); gpu.prof_enable(True)
    with _hold(gpu, hold):
        lg, al, _ = m.forward(video, cap, keep=keep, seed=seed)
        _, ga, ids = m.forward(video, None, greedy=True)
    import torch
    torch.cuda.synchronize()
    classes = {r["kernel_class"] for r in gpu.prof_collect()}; gpu.prof_enable(False)
    assert (9 in classes) != hold, classes                                         # the form this case is about really ran
    assert np.array_equal(al.cpu().numpy(), ref_a)
    assert np.array_equal(lg.cpu().numpy(), ref_l)
    _, ref_ga, ref_ids = oracle.attention_forward(p, d, video, None, greedy=True)
    assert np.array_equal(ids.cpu().numpy(), ref_ids) and np.array_equal(ga.cpu().numpy(), ref_ga)
    assert gpu.chain_timeouts() == 0


def test_attention32_benched_rows_gradients_vs_float64_autograd(gpu, oracle):
    """Gradients at the benched shape's rows, frames and width -- B = 64, Tv = 32, H = 1000 (|V| = 2000, Tc = 6 bound the float64 autograd
    pass): the backward recurrence's 20-frames-per-chunk path and attn_dpdv_kernel run at 64 rows.  Loss (regulariser live) and every gradient
    vs float64 autograd in the persistent form; the per-step form (`chain_hold`) gives the same update up to the order-free reductions."""
    import torch
    from s2vt_amd import hostglue
    D, V, H, Tv, Tc, B, keep = 1536, 2000, 1000, 32, 6, 64, 0.9
    d, p, m, video, cap, rng = _setup(oracle, D, V, H, Tv, Tc, B, 34)
    ln = 1 + np.minimum(rng.poisson(3, B), Tc - 2)
    for i in range(B):
        cap[i, ln[i]:] = 0
    mask = hostglue.masks_from_ids(cap)
    seed = m.dropout_seed
    ref_loss, ref_g, alphas = _ref_loss_grads(oracle, p, video, cap, mask, _drop(oracle, seed, B, H, Tc, keep), keep, m.beta, m.m)
    assert (alphas[:, :8, :].sum(1) < m.m).any(), "the test must exercise the open hinge"
    gpu.prof_filter(-1, -1); gpu.prof_enable(True)
    st = m.xe_update(video, cap, mask, lr=0.0, keep=keep, active_steps=None)
    torch.cuda.synchronize()
    classes = {r["kernel_class"] for r in gpu.prof_collect()}; gpu.prof_enable(False)
    assert 9 in classes and 10 in classes, classes
    assert abs(float(st.loss) - ref_loss) <= 1e-3 * max(1.0, abs(ref_loss))
    _check_grads(m, ref_g)
    gn = sum(float((g ** 2).sum()) for g in ref_g.values())
    assert abs(float(st.grad_sumsq) - gn) <= 1e-3 * gn
    g_pers = {k: m.store.g[k].clone() for k in m.store.names}
    m.global_step = 0
    with gpu.chain_hold():
        st2 = m.xe_update(video, cap, mask, lr=0.0, keep=keep, active_steps=None)
    assert abs(float(st2.loss) - float(st.loss)) <= 1e-6 * abs(float(st.loss))
    for k in m.store.names:
        a, b = m.store.g[k], g_pers[k]
        assert float((a - b).abs().max()) <= 3e-5 * float(b.abs().max()) + 1e-10, k
    assert gpu.chain_timeouts() == 0


def test_attention_replay_train_statements(gpu, oracle):
    """original_attention.py train() (:403-470): model, build_model, exponential_decay + Adam + clip 10 -> train_op, build_sampler,
    sess.run([train_op, tf_loss], feed_dict), sess.run(learning_rate), sess.run(greedy_captions, feed_dict); test() (:539):
    build_generator.  Feeds are host lists / numpy arrays as the reference's."""
    import torch
    from s2vt_amd import attention as A, hostglue
    from oracle import s2vt_torch as T
    D, V, H, Tv, Tc, B = 64, 140, 32, 5, 7, 4
    d, p, _, feats, _, rng = _setup(oracle, D, V, H, Tv, Tc, B, 6)
    vocabulary = ["<en_unk>"] + [f"w{i}" for i in range(V - 3)]
    wordtoix, ixtoword = hostglue.preProBuildWordVocab(vocabulary, word_count_threshold=0)
    model = A.Attention_Caption_Generator(dim_image=D, n_words=len(wordtoix), dim_hidden=H, batch_size=B, n_video_lstm_steps=Tv,
                                          n_caption_lstm_steps=Tc, drop_out_rate=0.9, bias_init_vector=None)
    model.load(p)
    tf_loss, tf_video, tf_caption, tf_caption_mask = model.build_model()
    sess = A.Session(model)
    learning_rate = model.exponential_decay(0.0001, 10000, 0.5)
    train_op = model.minimize((tf_loss, tf_video, tf_caption, tf_caption_mask), learning_rate, clip_norm=10)
    greedy_captions, greedy_video_features, saved_alphas = model.build_sampler()
    captions_batch = ["w1 w2 w3", "w4 w5 w6 w7 w8 w9 w10 w11 w12", "w3", "zzz w2"]
    captions_ind, captions_mask = hostglue.sentence_padding_toix(list(captions_batch), wordtoix, Tc)
    features_batch = [feats[j].tolist() for j in range(B)]
    cap = np.asarray(captions_ind, np.int32); mask = np.asarray(captions_mask, np.float32)
    drop = _drop(oracle, model.dropout_seed, B, H, Tc, 0.9)
    pt = T.to_torch(p, torch.float64, True)
    ref_loss, _, _ = T.attention_xe_loss(pt, torch.as_tensor(feats).double(), cap, mask, drop, 0.9, model.beta, model.m)
    ref_loss.backward()
    _, loss_val = sess.run([train_op, tf_loss], feed_dict={tf_video: features_batch, tf_caption: captions_ind, tf_caption_mask: captions_mask})
    assert abs(loss_val - float(ref_loss)) < 1e-5 * max(1.0, abs(float(ref_loss)))
    assert sess.run(learning_rate) == pytest.approx(0.0001) and model.global_step == 1
    # the update: tf.clip_by_global_norm(10) + TF-form Adam's first step, in float64 on the gradients the train_op left in the bucket
    g_gpu = {k: torch.as_tensor(model.store.g[k].cpu().numpy().astype(np.float64)) for k in model.store.names}
    for k in model.store.names:
        ref = pt[k].grad.numpy().reshape(model.store.shapes[k])
        assert np.abs(g_gpu[k].numpy() - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-9, k
    g, _ = T.clip_by_global_norm(g_gpu, 10.0)
    th = {k: torch.as_tensor(np.asarray(p[k], np.float64).reshape(model.store.shapes[k])) for k in model.store.names}
    th, _, _ = T.adam_tf(th, g, {k: torch.zeros_like(v) for k, v in th.items()}, {k: torch.zeros_like(v) for k, v in th.items()}, 1, 0.0001)
    for k in model.store.names:
        assert np.abs(model.store.p[k].cpu().numpy() - th[k].numpy()).max() <= 2e-3 * 0.0001 + 1e-7, k
    # greedy sampler on the updated variables vs the oracle on the same variables
    p2 = {k: model.store.p[k].cpu().numpy().reshape(np.shape(p[k])) for k in model.store.names}
    greedy_words, batch_alphas = sess.run([greedy_captions, saved_alphas], {greedy_video_features: features_batch})
    _, ref_a, ref_ids = oracle.attention_forward(p2, d, feats, None, greedy=True)
    assert greedy_words.dtype == np.int64 and np.array_equal(greedy_words, ref_ids) and np.array_equal(batch_alphas, ref_a)
    masks, decoded = hostglue.decode_captions_masks(np.array(greedy_words), ixtoword)
    assert len(decoded) == B and np.asarray(masks).shape == (B, Tc)
    video_tf, captions_tf = model.build_generator()
    gen = sess.run(captions_tf, feed_dict={video_tf: features_batch})
    assert np.array_equal(gen, ref_ids)
    # checkpoint names are the reference's TF variable names
    sd = model.state_dict()
    assert "s2vt/LSTM3/basic_lstm_cell/weights" in sd and "embed_att_Wa" in sd and sd["embed_att_w"].shape == (H, 1)


@pytest.mark.parametrize("D,V,H,Tv,Tc,B,keep", [
    (32, 101, 128, 5, 4, 32, 0.9),      # NG = 8 at its largest H; every workgroup of the forward grid is an attention workgroup
    (32, 101, 144, 3, 3, 20, 1.0),      # NG = 64 at its smallest H (9 k-groups of 64), partial last row tile
    (24, 67, 256, 6, 3, 64, 0.9),       # B = H / 4: query and attention roles overlap on 16 workgroups; two frame chunks
    (24, 67, 1024, 2, 2, 33, 0.9),      # every CU, no padded k-group, 3 row tiles (one of them a single row)
    (16, 53, 64, 1, 5, 1, 1.0),         # one row, one frame
    (16, 53, 1000, 11, 2, 16, 0.5),     # three frame chunks at the bench's H
])
def test_attention_persistent_recurrences_equal_per_step_launches(gpu, oracle, D, V, H, Tv, Tc, B, keep):
    """attn_chain.hip / attn_chain_bwd.hip against the per-step launches (ops.chain_hold() selects them): logits, alphas and EVERY saved
    activation the backward reads bit-identical; gradients equal to the noise of the order-free reductions.  Shapes chosen at the edges of
    the persistent forms (role overlap, k-group padding, partial row tiles, frame chunking)."""
    import torch
    from s2vt_amd import attention as A
    ops = gpu
    rng = np.random.default_rng(H + Tv)
    m = A.Attention_Caption_Generator(D, V, H, B, Tv, Tc, keep, seed=5, m=0.5 if Tv <= 8 else 0.9)
    for k in ("lstm3_b", "embed_att_ba", "embed_nn_bp", "embed_word_b", "encode_image_b"):
        m.p[k].copy_(torch.as_tensor(rng.uniform(-.1, .1, tuple(m.p[k].shape)).astype(np.float32)))
    video = torch.as_tensor(np.abs(rng.standard_normal((B, Tv, D)) * 0.5).astype(np.float32)).cuda()
    cap = rng.integers(0, V, (B, Tc)).astype(np.int32)
    mask = (rng.random((B, Tc)) < 0.8).astype(np.float32); mask[:, 0] = 1
    capd = torch.as_tensor(cap).cuda()
    vid, sid = m._row_ids(B)

    def run(hold):
        ops.prof_filter(-1, -1); ops.prof_enable(True)
        ctx = ops.chain_hold() if hold else None
        if ctx: ctx.__enter__()
        try:
            lg, al, ws = ops.attn_teacher_forced_fwd(m.dims, m.store.params, video, capd, keep, 99, vid, sid, want_alphas=True)
            lg, al, wsc = lg.clone(), al.clone(), ws.clone()
            m.global_step = 0
            st = m.xe_update(video, cap, mask, lr=0.0, keep=keep, active_steps=None)
            grads = {k: m.store.g[k].clone() for k in m.store.names}
            loss = float(st.loss)
        finally:
            if ctx: ctx.__exit__(None, None, None)
        torch.cuda.synchronize()
        classes = {r["kernel_class"] for r in ops.prof_collect()}; ops.prof_enable(False)
        return lg, al, wsc, grads, loss, classes
    pers = run(False)
    step = run(True)
    assert 9 in pers[5] and 10 in pers[5], pers[5]              # the persistent forms really ran ...
    assert 9 not in step[5] and 10 not in step[5], step[5]      # ... and did not under the hold
    assert torch.equal(pers[0], step[0]) and torch.equal(pers[1], step[1])
    # the saved activations: everything up to and including the output layer Y lies at the front of the workspace, in carve order
    n = lambda *d: (int(np.prod(d)) * 4 + 255) & ~255
    front = sum([n(Tv * B), n(Tc * B), n(Tc * B), n(B), n(B), n(Tv * B * H), n(Tv * B * H), n(Tc * B * H), n(Tc * Tv * B), n(Tc * B), n(Tc * B * H),
                 n(Tc * B * 4 * H), n((Tc + 1) * B * H), n((Tc + 1) * B * H), n((Tc + 1) * B * H), n(Tc * B * H)])
    skip = sum([n(Tv * B), n(Tc * B), n(Tc * B), n(B), n(B), n(Tv * B * H), n(Tv * B * H)]) + n(B * H)    # hWa slot 0 is never written (zero query)
    a, b = pers[2][:front], step[2][:front]
    hwa0 = sum([n(Tv * B), n(Tc * B), n(Tc * B), n(B), n(B), n(Tv * B * H), n(Tv * B * H)])
    assert torch.equal(a[:hwa0], b[:hwa0]) and torch.equal(a[hwa0 + B * H * 4:], b[hwa0 + B * H * 4:])
    assert abs(pers[4] - step[4]) <= 1e-6 * max(1.0, abs(step[4]))
    for k in m.store.names:
        ref = step[3][k]
        assert float((pers[3][k] - ref).abs().max()) <= 3e-5 * float(ref.abs().max()) + 1e-10, k
    assert ops.chain_timeouts() == 0
