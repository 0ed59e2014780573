"""GPU parity of the session API and the single-op entry points added for SURVEY section 8(b): the split
encode / decode calls reproduce s2vt_sample (and therefore the oracle) bit for bit; the single ops match
torch float64 restatements."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DIMS = dict(dim_image=128, n_words=260, word_dim=32, lstm_dim=64, n_video_lstm_step=5, n_caption_lstm_step=8)


def _dev(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def test_session_encode_decode_matches_oracle(gpu, oracle):
    import torch
    d = oracle.Dims(label_dim=0, **DIMS)
    p = oracle.init_params(d, seed=3)
    rng = np.random.default_rng(0)
    video = np.abs(rng.standard_normal((6, 5, 128)) * 0.5).astype(np.float32)
    ref_s, ref_g = oracle.sample_captions(p, d, video, K=3, seed=11)
    dims = gpu.make_dims(128, 260, 32, 64, 5, 8)
    dp = {k: _dev(v) for k, v in p.items()}
    params = gpu.make_params(dp)
    ses = gpu.Session(dims, max_B=8, max_K=4)
    ses.encode(params, _dev(video))
    g = ses.decode_greedy(params)
    s = ses.decode_multinomial(params, 3, seed=11)
    g2 = ses.decode_greedy(params)                       # a second decode on the same encode
    torch.cuda.synchronize()
    assert np.array_equal(g.cpu().numpy(), ref_g) and np.array_equal(g2.cpu().numpy(), ref_g)
    assert np.array_equal(s.cpu().numpy(), ref_s)
    ses.close()


def test_pack_unpack_weights_roundtrip(gpu):
    import torch
    E, H = 12, 20
    W = torch.randn(E + H, 4 * H, device="cuda")
    Wx, Wh = gpu.pack_weights(W, E)
    assert torch.equal(Wx[:, :, 2], W[:E, 2 * H:3 * H]) and torch.equal(Wh[:, :, 1], W[E:, H:2 * H])   # gate-interleaved views
    assert torch.equal(gpu.unpack_weights(Wx, Wh), W)


def test_embed_gather_and_clip(gpu):
    import torch
    W = torch.randn(97, 12, device="cuda")
    idx = torch.randint(0, 97, (33,), device="cuda", dtype=torch.int32)
    assert torch.equal(gpu.embed_gather(W, idx), W[idx.long()])
    g = torch.randn(5000, device="cuda") * 3
    g0 = g.clone().double()
    ss = gpu.global_norm_clip(g, 5.0)
    nrm = float(g0.norm())
    assert abs(float(ss) - nrm ** 2) <= 1e-4 * nrm ** 2
    assert torch.allclose(g.double(), g0 * (5.0 / max(nrm, 5.0)), rtol=1e-5, atol=1e-7)


def test_lstm_cell_bwd_and_frame_embed_bwd_vs_autograd(gpu):
    import torch
    torch.manual_seed(0)
    M, E, H = 7, 12, 20
    W = (torch.randn(E + H, 4 * H, dtype=torch.float64) * 0.3).requires_grad_()
    b = torch.randn(4 * H, dtype=torch.float64) * 0.1
    x = torch.randn(M, E, dtype=torch.float64); h = torch.randn(M, H, dtype=torch.float64)
    c = torch.randn(M, H, dtype=torch.float64).requires_grad_()
    z = (torch.cat([x, h], 1) @ W + b).requires_grad_()
    i, j, f, o = z.split(H, 1)
    c_new = c * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
    h_new = torch.tanh(c_new) * torch.sigmoid(o)
    dh = torch.randn(M, H, dtype=torch.float64); dc_in = torch.randn(M, H, dtype=torch.float64)
    dz_ref, dc_ref = torch.autograd.grad((h_new * dh).sum() + (c_new * dc_in).sum(), [z, c])
    f32 = lambda t: t.detach().float().cuda().contiguous()
    gc, gh, _, gates = gpu.lstm_cell_fwd(gpu.operand(f32(x)), None, f32(h), f32(c), f32(W), f32(b), M, want_gates=True)
    dz, dc_prev = gpu.lstm_cell_bwd(gates, gc, f32(c), f32(dh), f32(dc_in))
    assert np.allclose(dz.cpu().numpy(), dz_ref.numpy(), rtol=2e-4, atol=2e-5)
    assert np.allclose(dc_prev.cpu().numpy(), dc_ref.numpy(), rtol=2e-4, atol=2e-5)
    # frame embedding backward
    dims = gpu.make_dims(24, 97, E, H, 3, 6)
    video = torch.randn(4, 3, 24, dtype=torch.float64); demb = torch.randn(12, E, dtype=torch.float64)
    dW = torch.zeros(24, E, device="cuda"); db = torch.zeros(E, device="cuda")
    gpu.frame_embed_bwd(dims, f32(video), f32(demb), dW, db)
    assert np.allclose(dW.cpu().numpy(), (video.view(12, 24).t() @ demb).numpy(), rtol=1e-4, atol=1e-4)
    assert np.allclose(db.cpu().numpy(), demb.sum(0).numpy(), rtol=1e-4, atol=1e-4)


def test_named_losses(gpu, oracle):
    import torch
    rng = np.random.default_rng(2)
    N, Tc, V = 6, 5, 131
    logits = (2 * rng.standard_normal((Tc * N, V))).astype(np.float32)
    tgt = rng.integers(0, V, Tc * N).astype(np.int32)
    adv = rng.standard_normal(N).astype(np.float32); mask = (rng.random((N, Tc)) > 0.3).astype(np.float32)
    nll_ref, lp_ref, _ = oracle.row_losses(logits, tgt, 0.0)
    dl = _dev(logits)
    nll, lp, coef = gpu.pg_nll_fwd_bwd(dl, _dev(tgt), _dev(adv), _dev(mask))
    assert np.allclose(coef.cpu().numpy(), (mask * adv[:, None]).T.reshape(-1))
    assert np.allclose(nll.cpu().numpy(), nll_ref, rtol=1e-5, atol=1e-5) and np.allclose(lp.cpu().numpy(), lp_ref, rtol=1e-5, atol=1e-5)
    lt = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
    (-(torch.log_softmax(lt, -1)[torch.arange(Tc * N), torch.as_tensor(tgt).long()]) * torch.tensor((mask * adv[:, None]).T.reshape(-1))).sum().backward()
    assert np.allclose(dl.cpu().numpy(), lt.grad.numpy(), rtol=1e-4, atol=1e-6)
    nll_s_ref, _, _ = oracle.row_losses(logits, tgt, 0.05)
    nll_s = gpu.xent_smooth_fwd_bwd(_dev(logits), _dev(tgt), torch.ones(Tc * N, device="cuda"), 0.05)
    assert np.allclose(nll_s.cpu().numpy(), nll_s_ref, rtol=1e-5, atol=1e-5)
