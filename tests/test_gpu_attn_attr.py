"""GPU parity: temporal attention (original_attention.py:95-134) and the attribute head
(reinforce_multitask_e2e_attribute_loss.py:375-380): forward bit-exact vs the C oracle, backward vs
float64 autograd."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("Tv,B,H", [(5, 3, 8), (5, 64, 1000), (7, 20, 36)])
def test_attention_step_bit_exact_and_backward(gpu, oracle, Tv, B, H):
    import torch
    rng = np.random.default_rng(B)
    hWa = rng.standard_normal((B, H)).astype(np.float32); P = rng.standard_normal((Tv, B, H)).astype(np.float32)
    Vt = rng.standard_normal((Tv, B, H)).astype(np.float32); w = rng.uniform(-.1, .1, H).astype(np.float32)
    alpha, ctx = oracle.attention_step(hWa, P, Vt, w)
    sc, al, cx = gpu.attention_fwd(_dev(hWa), _dev(P), _dev(Vt), _dev(w))
    assert np.array_equal(al.cpu().numpy(), alpha) and np.array_equal(cx.cpu().numpy(), ctx)
    # backward vs autograd (float64)
    t = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    th, tP, tV, tw = t(hWa), t(P), t(Vt), t(w)
    e = (torch.tanh(th + tP) * tw).sum(-1)
    a = torch.exp(e) / torch.exp(e).sum(0)
    c = (a.unsqueeze(-1) * tV).sum(0)
    dctx = rng.standard_normal((B, H)).astype(np.float32)
    (c * torch.tensor(dctx, dtype=torch.float64)).sum().backward()
    dw = torch.zeros(H, device="cuda")
    dh, dP, dV = gpu.attention_bwd(_dev(hWa), _dev(P), _dev(Vt), _dev(w), al, _dev(dctx), dw)
    for got, ref in ((dh, th.grad), (dP, tP.grad), (dV, tV.grad), (dw, tw.grad)):
        ref = ref.numpy()
        assert np.abs(got.cpu().numpy() - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-7


def test_attention_model_forward_bit_exact(gpu, oracle):
    """Whole teacher-forced attention graph and the greedy generator, vs oracle.attention_forward."""
    import s2vt_amd
    from s2vt_amd import attention as A
    d = oracle.Dims(dim_image=48, n_words=131, word_dim=0, lstm_dim=32, n_video_lstm_step=5, n_caption_lstm_step=6, label_dim=0)
    p = oracle.init_attention_params(d, 3)
    rng = np.random.default_rng(2)
    for k in ("lstm3_b", "embed_att_ba", "embed_nn_bp", "embed_word_b", "encode_image_b"):
        p[k] = rng.uniform(-.1, .1, p[k].shape).astype(np.float32)
    B = 5
    video = np.abs(rng.standard_normal((B, 5, 48))).astype(np.float32)
    cap = rng.integers(0, 131, (B, 6)).astype(np.int32)
    m = A.Attention_Caption_Generator(48, 131, 32, B, 5, 6, 1.0)
    m.load(p)
    ref_l, ref_a, _ = oracle.attention_forward(p, d, video, cap)
    lg, al, _ = m.forward(video, cap)
    assert np.array_equal(lg.cpu().numpy(), ref_l) and np.array_equal(al.cpu().numpy(), ref_a)
    _, _, ref_ids = oracle.attention_forward(p, d, video, None, greedy=True)
    _, _, ids = m.forward(video, None, greedy=True)
    assert np.array_equal(ids.cpu().numpy(), ref_ids)


@pytest.mark.parametrize("B,Tv,D,A", [(3, 5, 6, 4), (64, 5, 1536, 400)])
def test_attr_head_bit_exact_and_backward(gpu, oracle, B, Tv, D, A):
    import torch
    rng = np.random.default_rng(A)
    p = {"attr_W": rng.uniform(-.1, .1, (D, A)).astype(np.float32), "attr_b": rng.uniform(-.1, .1, A).astype(np.float32)}
    video = np.abs(rng.standard_normal((B, Tv, D))).astype(np.float32); y = (rng.random((B, A)) < .2).astype(np.float32)
    z_ref, bce_ref = oracle.attr_head(p, video, y)
    mean, z, bce = gpu.attr_head_fwd(_dev(video), _dev(p["attr_W"]), _dev(p["attr_b"]), _dev(y))
    assert np.array_equal(z.cpu().numpy(), z_ref) and np.array_equal(bce.cpu().numpy(), bce_ref)
    from oracle import s2vt_torch as T
    pt = T.to_torch(p, torch.float64, True)
    T.attr_bce(pt, torch.as_tensor(video).double(), y, normalise=True).backward()
    dW = torch.zeros(D, A, device="cuda"); db = torch.zeros(A, device="cuda")
    gpu.attr_head_bwd(mean, z, _dev(y), 1.0 / (A * B), dW, db)
    for got, ref in ((dW, pt["attr_W"].grad), (db, pt["attr_b"].grad)):
        ref = ref.numpy()
        assert np.abs(got.cpu().numpy() - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-9


def test_multitask_reinforce_gradients(gpu, oracle):
    """-(1-alpha)*PG/sum(mask) + alpha*sum(bce)/(A*B)  (reinforce_multitask_e2e_attribute_loss.py:957)."""
    import torch
    import s2vt_amd
    from s2vt_amd import model as M
    from oracle import s2vt_torch as T
    d = oracle.Dims(dim_image=24, n_words=97, word_dim=12, lstm_dim=20, n_video_lstm_step=3, n_caption_lstm_step=6, label_dim=10)
    p = oracle.init_params(d, seed=3, attr=True)
    rng = np.random.default_rng(4)
    B, rep, alpha, keep = 4, 1, 0.05, 0.9
    video = np.abs(rng.standard_normal((B, 3, 24)) * 0.5).astype(np.float32)
    cap = rng.integers(0, 97, (B, 6)).astype(np.int32); cap[:, -1] = 0
    mask = s2vt_amd.hostglue.masks_from_ids(cap)
    r = rng.random(B).astype(np.float32); b = rng.random(B).astype(np.float32)
    y = (rng.random((B, 10)) < .3).astype(np.float32)
    mdl = M.Video_Caption_Generator(24, 97, 12, 20, B, 0, 3, 6, dropout_rate=keep, label_dim=10, alpha=alpha)
    mdl.store.load(p)
    vid = np.arange(B, dtype=np.int32); sid = np.zeros(B, np.int32)
    drop = oracle.dropout_masks(mdl.dropout_seed, vid, sid, keep, 20, 3, 6)
    pt = T.to_torch(p, torch.float64, True)
    logits = T.teacher_forced(pt, torch.as_tensor(video).double(), cap, drop, keep)
    loss = (1 - alpha) * T.pg_loss(logits, cap, mask, r, b) + alpha * T.attr_bce(pt, torch.as_tensor(video).double(), y)
    loss.backward()
    st = mdl.reinforce_update(video, cap, mask, r, b, lr=0.0, clip_norm=10.0, true_labels=y)
    assert abs(float(st.loss) + float(st.attr_loss) - float(loss)) < 1e-5
    for n in mdl.store.names:
        ref = pt[n].grad.numpy()
        got = mdl.store.g[n].cpu().numpy()
        assert np.abs(got - ref).max() <= 2e-4 * (np.abs(ref).max() + 1e-12) + 1e-9, n


def test_attention_training_gradients_vs_float64_autograd(gpu, oracle):
    """The attention captioner's training step: loss and d(loss)/d(every variable) composed from the library's
    backward pieces vs float64 autograd of oracle/s2vt_torch.py::attention_teacher_forced, with the same
    Philox dropout masks (the dropped LSTM3 output also feeds the next step's attention query, :135)."""
    import torch
    import s2vt_amd
    from s2vt_amd import attention as A
    from oracle import s2vt_torch as T
    d = oracle.Dims(dim_image=48, n_words=131, word_dim=0, lstm_dim=32, n_video_lstm_step=5, n_caption_lstm_step=6, label_dim=0)
    p = oracle.init_attention_params(d, 3)
    rng = np.random.default_rng(4)
    for k in ("lstm3_b", "embed_att_ba", "embed_nn_bp", "embed_word_b", "encode_image_b"):
        p[k] = rng.uniform(-.1, .1, p[k].shape).astype(np.float32)
    B, keep, seed = 5, 0.9, 321
    video = np.abs(rng.standard_normal((B, 5, 48))).astype(np.float32)
    cap = rng.integers(0, 131, (B, 6)).astype(np.int32)
    mask = (rng.random((B, 6)) < 0.8).astype(np.float32); mask[:, 0] = 1
    vid = np.arange(B, dtype=np.int32); sid = np.zeros(B, np.int32)
    drop = [oracle.dropout_mask(seed, vid, sid, 768 + t, keep, 32) for t in range(6)]
    pt = T.to_torch(p, torch.float64, True)
    logits, _ = T.attention_teacher_forced(pt, torch.as_tensor(video).double(), cap, drop, keep)
    lp = torch.log_softmax(logits, -1)
    ce = -lp.gather(2, torch.as_tensor(cap).long().unsqueeze(-1)).squeeze(-1)
    ref_loss = (ce * torch.as_tensor(mask).double()).sum() / float(mask.sum())
    ref_loss.backward()
    m = A.Attention_Caption_Generator(48, 131, 32, B, 5, 6, keep)
    m.load(p)
    loss, g = m.xe_update(video, cap, mask, lr=0.0, keep=keep, seed=seed)
    assert abs(float(loss) - float(ref_loss)) < 1e-5 * max(1.0, abs(float(ref_loss)))
    for k in g:
        ref = pt[k].grad.numpy()
        got = g[k].cpu().numpy().astype(np.float64)
        assert np.abs(got - ref).max() <= 2e-4 * (np.abs(ref).max() + 1e-12) + 1e-9, k
    # and it trains: a few Adam steps lower the loss on the same batch
    l0 = float(m.xe_update(video, cap, mask, lr=1e-2, keep=1.0)[0])
    for _ in range(8):
        l1 = float(m.xe_update(video, cap, mask, lr=1e-2, keep=1.0)[0])
    assert l1 < l0
