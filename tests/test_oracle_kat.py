"""Known-answer tests that pin the CPU oracle: published Philox vectors, libm-accurate
transcendental functions, and independent float64 numpy restatements at toy dims."""
import numpy as np
import pytest


def test_philox_published_vectors(oracle):
    # Random123 kat_vectors, philox4x32 with 10 rounds
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, exp in kat:
        assert [int(x) for x in oracle.philox4x32_10(ctr, key)] == exp


def test_detmath_accuracy(oracle):
    x = np.linspace(-20, 20, 400001).astype(np.float32)
    assert np.abs(oracle.det_tanh(x) - np.tanh(x.astype(np.float64))).max() < 5e-7
    x = np.linspace(-87, 87, 400001).astype(np.float32)
    ref = np.exp(x.astype(np.float64))
    assert (np.abs(oracle.det_exp(x) - ref) / ref).max() < 2e-7
    x = np.exp(np.linspace(-80, 80, 400001)).astype(np.float32)
    ref = np.log(x.astype(np.float64))
    assert (np.abs(oracle.det_log(x) - ref) / np.maximum(np.abs(ref), 1e-3)).max() < 2e-7
    x = np.linspace(-100, 100, 400001).astype(np.float32)
    s = oracle.det_sigmoid(x)
    assert np.abs(s - 1 / (1 + np.exp(-x.astype(np.float64)))).max() < 2e-7
    assert s.min() >= np.finfo(np.float32).tiny      # never denormal
    assert np.all(np.diff(s) >= 0)


def test_gumbel_noise_statistics(oracle):
    g = np.concatenate([oracle.gumbel_noise(99, v, 0, 3, 12000) for v in range(20)]).astype(np.float64)
    assert abs(g.mean() - 0.5772) < 0.01 and abs(g.var() - np.pi ** 2 / 6) < 0.03
    # streams differ with every counter field
    a = oracle.gumbel_noise(99, 0, 0, 3, 64)
    for other in (oracle.gumbel_noise(98, 0, 0, 3, 64), oracle.gumbel_noise(99, 1, 0, 3, 64),
                  oracle.gumbel_noise(99, 0, 1, 3, 64), oracle.gumbel_noise(99, 0, 0, 4, 64)):
        assert not np.array_equal(a, other)


def test_gemm_chain_matches_float64(oracle):
    rng = np.random.default_rng(0)
    for M, K, N in [(1, 1, 1), (3, 7, 5), (17, 33, 65), (5, 300, 259)]:
        A = rng.standard_normal((M, K)).astype(np.float32)
        W = rng.standard_normal((K, N)).astype(np.float32)
        C = oracle.gemm_chain(A, W)
        ref = A.astype(np.float64) @ W.astype(np.float64)
        bound = 2e-7 * K * (np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64)) + 1e-7
        assert np.all(np.abs(C - ref) <= bound)
    # the chain is literally sequential fmaf: check against a python-loop float32 fma emulation
    A = rng.standard_normal((2, 9)).astype(np.float32); W = rng.standard_normal((9, 3)).astype(np.float32)
    C = oracle.gemm_chain(A, W)
    for m in range(2):
        for n in range(3):
            acc = np.float32(0)
            for k in range(9):
                acc = np.float32(np.float64(A[m, k]) * np.float64(W[k, n]) + np.float64(acc))  # exact product, one rounding
            assert acc == C[m, n]


def test_gemm_chain_segments_and_gather(oracle):
    rng = np.random.default_rng(1)
    A0 = rng.standard_normal((6, 5)).astype(np.float32); T = rng.standard_normal((11, 4)).astype(np.float32)
    idx = np.array([3, 3, 0, 10, 7, 1], np.int32)
    W = rng.standard_normal((9, 8)).astype(np.float32)
    C = oracle.gemm_chain(A0, W[:5]); oracle.gemm_chain(T, W[5:], C, rowidx=idx)
    full = np.concatenate([A0, T[idx]], 1)
    assert np.array_equal(C, oracle.gemm_chain(full, W))      # continuing a chain == one chain


def _np_lstm(x, c, h, W, b):
    z = np.concatenate([x, h], 1).astype(np.float64) @ W.astype(np.float64) + b
    H = h.shape[1]
    i, j, f, o = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]
    sig = lambda v: 1 / (1 + np.exp(-v))
    c2 = c * sig(f + 1.0) + sig(i) * np.tanh(j)
    return c2, np.tanh(c2) * sig(o)


def test_lstm_cell_known_answer(oracle):
    """BasicLSTMCell semantics: gate order i,j,f,o; forget_bias 1.0; rows [x;h]."""
    rng = np.random.default_rng(2)
    M, E, H = 5, 3, 4
    p = {"lstm1_W": rng.uniform(-1, 1, (E + H, 4 * H)).astype(np.float32), "lstm1_b": rng.uniform(-.5, .5, 4 * H).astype(np.float32)}
    x = rng.standard_normal((M, E)).astype(np.float32); c = rng.standard_normal((M, H)).astype(np.float32)
    h = rng.uniform(-1, 1, (M, H)).astype(np.float32)
    c2, h2, out, gates, _ = oracle.lstm1_step(p, x, c, h, want_gates=True)
    rc, rh = _np_lstm(x, c, h, p["lstm1_W"], p["lstm1_b"])
    assert np.abs(c2 - rc).max() < 2e-6 and np.abs(h2 - rh).max() < 2e-6
    assert np.array_equal(out, h2)
    # hand-checkable case: zero weights -> z = b
    p0 = {"lstm1_W": np.zeros((E + H, 4 * H), np.float32), "lstm1_b": np.zeros(4 * H, np.float32)}
    c2, h2, _, _, _ = oracle.lstm1_step(p0, x, np.ones((M, H), np.float32), h)
    sig1 = 1 / (1 + np.exp(-1.0))
    assert np.allclose(c2, sig1, atol=1e-6)           # c*sig(0+1) + sig(0)*tanh(0)
    assert np.allclose(h2, np.tanh(sig1) * 0.5, atol=1e-6)
    # zero input == absent input (decode-stage padding)
    a = oracle.lstm1_step(p, np.zeros((M, E), np.float32), c, h)
    b = oracle.lstm1_step(p, None, c, h)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_dropout_wrapper(oracle):
    rng = np.random.default_rng(3)
    M, H = 4, 6
    z = rng.standard_normal((M, 4 * H)).astype(np.float32); c = rng.standard_normal((M, H)).astype(np.float32)
    mask = (rng.random((M, H)) < 0.9).astype(np.float32)
    c2, h2, out, _ = oracle.lstm_pointwise(z, c, mask, 0.9)
    assert np.array_equal(out, (h2 / np.float32(0.9)) * mask)   # state not dropped, output scaled by 1/keep
    assert np.array_equal(oracle.lstm_pointwise(z, c)[1], h2)


def test_pick_tokens(oracle):
    rng = np.random.default_rng(4)
    M, V = 6, 37
    logits = rng.standard_normal((M, V)).astype(np.float32)
    logits[0, 5] = logits[0, 9] = 10.0                          # tie -> lowest index (tf.argmax)
    vid = np.arange(M, dtype=np.int32); greedy = -np.ones(M, np.int32)
    tok = oracle.pick_tokens(logits, vid, greedy, 0, 7)
    assert tok[0] == 5 and np.array_equal(tok[1:], logits[1:].argmax(1))
    sid = np.zeros(M, np.int32)
    tok = oracle.pick_tokens(logits, vid, sid, 2, 7)
    for m in range(M):
        g = oracle.gumbel_noise(7, m, 0, 2, V)
        assert tok[m] == int(np.argmax(logits[m] + g))


def test_sampler_draws_from_softmax(oracle):
    """Chi-square: Gumbel-max draws follow softmax(logits) (what tf.multinomial promises)."""
    V, n = 8, 4000
    logits = np.array([[0.0, 1.0, -1.0, 2.0, 0.5, -2.0, 1.5, 0.2]], np.float32)
    pr = np.exp(logits[0].astype(np.float64)); pr /= pr.sum()
    L = np.repeat(logits, n, 0)
    tok = oracle.pick_tokens(L, np.arange(n, dtype=np.int32), np.zeros(n, np.int32), 0, 12345)
    cnt = np.bincount(tok, minlength=V)
    chi2 = ((cnt - n * pr) ** 2 / (n * pr)).sum()
    assert chi2 < 24.3          # chi2(7 dof) 99.9th percentile


def test_row_losses(oracle):
    rng = np.random.default_rng(5)
    M, V = 7, 23
    logits = (3 * rng.standard_normal((M, V))).astype(np.float32)
    tgt = rng.integers(0, V, M).astype(np.int32)
    lp_ref = logits.astype(np.float64) - np.log(np.exp(logits.astype(np.float64)).sum(1, keepdims=True))
    nll, lp, lse = oracle.row_losses(logits, tgt, 0.0)
    assert np.allclose(lp, lp_ref[np.arange(M), tgt], atol=2e-6) and np.allclose(nll, -lp)
    nll, _, _ = oracle.row_losses(logits, tgt, 0.05)
    q = np.full((M, V), 0.05 / V); q[np.arange(M), tgt] += 0.95
    assert np.allclose(nll, -(q * lp_ref).sum(1), atol=5e-6)


def test_xe_loss_q1_semantics(oracle):
    """Q1: TF-1.1 softmax_cross_entropy returns the batch mean, multiplied by the mask column."""
    rng = np.random.default_rng(6)
    d = oracle.Dims(dim_image=6, n_words=11, word_dim=3, lstm_dim=4, n_video_lstm_step=2, n_caption_lstm_step=3)
    p = oracle.init_params(d, 1)
    N = 4
    logits = rng.standard_normal((N, 3, 11)).astype(np.float32)
    cap = rng.integers(0, 11, (N, 3)).astype(np.int32)
    mask = np.array([[1, 1, 0], [1, 0, 0], [1, 1, 1], [1, 1, 0]], np.float32)
    lp = logits.astype(np.float64) - np.log(np.exp(logits.astype(np.float64)).sum(-1, keepdims=True))
    q = np.full(logits.shape, 0.05 / 11); np.put_along_axis(q, cap[..., None].astype(np.int64), 0.95 + 0.05 / 11, -1)
    ce = -(q * lp).sum(-1)
    wd = sum(0.5 * (v.astype(np.float64) ** 2).sum() for k, v in p.items() if k not in ("lstm1_b", "lstm2_b"))
    exp_q1 = (ce.mean(0, keepdims=True) * mask).sum() / mask.sum() + 5e-5 * wd
    exp_plain = (ce * mask).sum() / mask.sum() + 5e-5 * wd
    assert abs(oracle.xe_loss(p, d, logits, cap, mask) - exp_q1) < 1e-6
    assert abs(oracle.xe_loss(p, d, logits, cap, mask, q1=False) - exp_plain) < 1e-6
    assert abs(exp_q1 - exp_plain) > 1e-4


def test_attention_step_and_attr_head(oracle):
    rng = np.random.default_rng(7)
    Tv, B, H = 5, 3, 8
    hWa = rng.standard_normal((B, H)).astype(np.float32); P = rng.standard_normal((Tv, B, H)).astype(np.float32)
    Vt = rng.standard_normal((Tv, B, H)).astype(np.float32); w = rng.standard_normal(H).astype(np.float32)
    alpha, ctx = oracle.attention_step(hWa, P, Vt, w)
    e = (np.tanh(hWa[None].astype(np.float64) + P) * w).sum(-1)
    a = np.exp(e) / np.exp(e).sum(0)
    assert np.allclose(alpha, a, atol=1e-6) and np.allclose(ctx, (a[..., None] * Vt).sum(0), atol=1e-5)
    D, A = 6, 4
    p = {"attr_W": rng.standard_normal((D, A)).astype(np.float32), "attr_b": rng.standard_normal(A).astype(np.float32)}
    video = rng.random((B, Tv, D)).astype(np.float32); y = (rng.random((B, A)) < .3).astype(np.float32)
    z, bce = oracle.attr_head(p, video, y)
    zr = video.astype(np.float64).mean(1) @ p["attr_W"] + p["attr_b"]
    ref = np.maximum(zr, 0) - zr * y + np.log1p(np.exp(-np.abs(zr)))
    assert np.allclose(z, zr, atol=1e-5) and np.allclose(bce, ref, atol=1e-5)


def test_oracle_vs_torch_restatement(oracle):
    """The bit-exact C oracle and the differentiable torch restatement are the same graph."""
    import torch
    from oracle import s2vt_torch as T
    d = oracle.Dims(dim_image=10, n_words=29, word_dim=6, lstm_dim=8, n_video_lstm_step=3, n_caption_lstm_step=4)
    p = oracle.init_params(d, 5)
    for k in ("lstm1_b", "lstm2_b", "encode_image_b", "embed_word_b"):
        p[k] = np.random.default_rng(1).uniform(-.2, .2, p[k].shape).astype(np.float32)
    rng = np.random.default_rng(8)
    N = 5
    video = np.abs(rng.standard_normal((N, 3, 10))).astype(np.float32)
    cap = rng.integers(0, 29, (N, 4)).astype(np.int32)
    H = 8
    drop = {k: (rng.random((T_, N, H)) < 0.9).astype(np.float32) for k, T_ in (("enc1", 3), ("enc2", 3), ("dec1", 4), ("dec2", 4))}
    lo = oracle.teacher_forced(p, d, video, cap, drop, 0.9)
    pt = T.to_torch(p, torch.float64, False)
    lt = T.teacher_forced(pt, torch.as_tensor(video).double(), cap, drop, 0.9).numpy()
    assert np.abs(lo - lt).max() < 5e-6
    mask = (rng.random((N, 4)) < .7).astype(np.float32); mask[:, 0] = 1
    r = rng.random(N); b = rng.random(N)
    assert abs(oracle.pg_loss(lo, cap, mask, r, b) - float(T.pg_loss(torch.as_tensor(lt), cap, mask, r, b))) < 1e-6
    assert abs(oracle.xe_loss(p, d, lo, cap, mask) - float(T.xe_loss(pt, torch.as_tensor(lt), cap, mask))) < 1e-6
