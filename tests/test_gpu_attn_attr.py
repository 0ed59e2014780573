"""GPU parity: temporal attention (original_attention.py:95-134) and the attribute head
(reinforce_multitask_e2e_attribute_loss.py:375-380): forward bit-exact vs the C oracle, backward vs
float64 autograd."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("Tv,B,H", [(5, 3, 8), (5, 64, 1000), (7, 20, 36), (32, 16, 1000), (50, 3, 24), (6, 5, 30)])
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


@pytest.mark.parametrize("B,Tv,D,A", [(4, 5, 96, 40), (32, 5, 1536, 400), (3, 10, 130, 7)])
def test_evaluate_multilabel_scores_bit_exact(gpu, oracle, B, Tv, D, A):
    """evaluate_multilabel (reinforce_multitask_e2e_attribute_loss.py:606-626): scores = sigmoid(mean_t(video) . attr_W + attr_b),
    through the C ABI (s2vt_attr_head_scores), the drop-in class (evaluate_multilabel -> Session.run) and the CNN wrapper (dropout off)."""
    import torch
    from s2vt_amd import e2e, model as M
    rng = np.random.default_rng(B * 7 + A)
    video = np.abs(rng.standard_normal((B, Tv, D)) * 0.5).astype(np.float32)
    p = {"attr_W": rng.uniform(-.1, .1, (D, A)).astype(np.float32), "attr_b": rng.uniform(-.5, .5, A).astype(np.float32)}
    ref = oracle.attr_scores(p, video)
    assert ref.shape == (B, A) and ref.min() > 0 and ref.max() < 1 and abs(float(ref.mean()) - 0.5) < 0.2
    z, sc = gpu.attr_head_scores(_dev(video), _dev(p["attr_W"]), _dev(p["attr_b"]))
    assert np.array_equal(z.cpu().numpy(), oracle.attr_head(p, video)[0])
    assert np.array_equal(sc.cpu().numpy(), ref)
    mdl = M.Video_Caption_Generator(D, 50, 8, 16, B, 0, Tv, 4, label_dim=A, alpha=0.05)
    mdl.store.load(p)
    tf_video, tf_scores = mdl.evaluate_multilabel(0.5)
    got = M.Session(mdl).run(tf_scores, feed_dict={tf_video: video})
    assert got.shape == (B, A) and np.array_equal(got, ref)
    assert np.array_equal(got > 0.5, oracle.attr_head(p, video)[0] > 0)             # the caller's threshold (sigmoid(z) > .5 <=> z > 0)
    # through a CNN (e2e.EndToEnd.evaluate_multilabel): inference mode -- no feature dropout (:613-620) -- so frames whose
    # "CNN" is the identity on [n, D] give the same scores
    class Ident(torch.nn.Module):
        def __init__(self):
            super().__init__(); self.w = torch.nn.Parameter(torch.ones(1))
        def forward(self, x):
            return x.reshape(x.shape[0], -1) * self.w
    tr = e2e.EndToEnd(mdl, Ident(), feature_keep=0.5)
    got2 = tr.evaluate_multilabel(torch.as_tensor(video).reshape(B, Tv, D, 1, 1)).cpu().numpy()
    assert np.array_equal(got2, ref)
    mdl0 = M.Video_Caption_Generator(D, 50, 8, 16, B, 0, Tv, 4)
    with pytest.raises(ValueError):
        mdl0.attribute_scores(video)
