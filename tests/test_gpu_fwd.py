"""GPU parity, forward path: HIP kernels through the C ABI vs the CPU oracle -- BIT-EXACT.

Every comparison is np.array_equal on fp32 values (== treats -0 and +0 alike) or on token ids.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(a, dtype=None):
    import torch
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def test_math_bitwise(gpu, oracle):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(500000) * 30, np.linspace(-100, 100, 200001), [0.0, -0.0, 87, -87, 9, -9, 1e-30]])
    x = x.astype(np.float32)
    for fn, ref in (("exp", oracle.det_exp), ("tanh", oracle.det_tanh), ("sigmoid", oracle.det_sigmoid)):
        y = gpu.math_eval(fn, _dev(x)).cpu().numpy()
        assert np.array_equal(y.view(np.uint32), ref(x).view(np.uint32)), fn
    xp = np.abs(x[np.abs(x) > 1e-37]) + np.float32(1e-37)
    xp = np.concatenate([xp, np.exp(rng.uniform(-80, 80, 300000)).astype(np.float32)])
    y = gpu.math_eval("log", _dev(xp)).cpu().numpy()
    assert np.array_equal(y.view(np.uint32), oracle.det_log(xp).view(np.uint32))


def test_gumbel_stream_bitwise(gpu, oracle):
    for seed, v, s, t, V in [(2024, 0, 0, 0, 12000), (2**40 + 17, 63, 4, 19, 9972), (1, 1000000, 7, 3, 101)]:
        g = gpu.gumbel_eval(seed, v, s, t, V).cpu().numpy()
        assert np.array_equal(g.view(np.uint32), oracle.gumbel_noise(seed, v, s, t, V).view(np.uint32))


SHAPES = [  # M, K, N
    (1, 1, 1), (3, 7, 5), (17, 33, 65), (64, 32, 64), (65, 100, 130), (5, 300, 259), (130, 96, 48), (200, 36, 20),
    (200, 1000, 388),                                  # several row / column tiles of every configuration, vector path, ragged edges
]


@pytest.mark.parametrize("M,K,N", SHAPES)
def test_gemm_bitwise_all_tiles(gpu, oracle, M, K, N):
    import s2vt_amd
    rng = np.random.default_rng(M * 1000 + K)
    A = rng.standard_normal((M, K)).astype(np.float32); W = rng.standard_normal((K, N)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    ref = oracle.bias_add(oracle.gemm_chain(A, W), b)
    dA, dW, db = _dev(A), _dev(W), _dev(b)
    for cfg in range(-1, 12):                           # every entry of fwd.hip kStore (64x96, 96x96, 96x128; 8-11: the LDS-DMA ring tiles)
        C = gpu.gemm([gpu.operand(dA)], dW, db, M=M, tile_cfg=cfg).cpu().numpy()
        assert np.array_equal(C, ref), f"tile cfg {cfg}"
    ref_t = oracle.det_tanh(ref)
    C = gpu.gemm([gpu.operand(dA)], dW, db, M=M, act_tanh=True).cpu().numpy()
    assert np.array_equal(C, ref_t)


def test_gemm_segments_gather_broadcast_cinit(gpu, oracle):
    rng = np.random.default_rng(11)
    for (M, k0, k1, k2, N, T, mod) in [(10, 8, 4, 12, 16, 9, 5), (70, 100, 36, 64, 132, 50, 7), (33, 5, 3, 6, 7, 4, 3)]:
        A0 = rng.standard_normal((mod, k0)).astype(np.float32)          # broadcast rows m % mod
        Tab = rng.standard_normal((T, k1)).astype(np.float32)           # gathered rows
        idx = rng.integers(0, T, M).astype(np.int32)
        A2 = rng.standard_normal((M, k2)).astype(np.float32)
        W = rng.standard_normal((k0 + k1 + k2 + 3, N)).astype(np.float32)[:k0 + k1 + k2]
        Ci = rng.standard_normal((M, N)).astype(np.float32)
        ref = Ci.copy()
        oracle.gemm_chain(np.ascontiguousarray(A0[np.arange(M) % mod]), W[:k0], ref)
        oracle.gemm_chain(Tab, W[k0:k0 + k1], ref, rowidx=idx)
        oracle.gemm_chain(A2, W[k0 + k1:], ref)
        dW = _dev(W)
        segs = [gpu.operand(_dev(A0), rowmod=mod), gpu.operand(_dev(Tab), rowidx=_dev(idx)), gpu.operand(_dev(A2))]
        for cfg in (-1, 0, 2, 8, 9):                   # (8, 9: LDS-DMA ring -- gathered / broadcast rows through the DMA source offsets)
            C = gpu.gemm(segs, dW, None, M=M, cinit=_dev(Ci), tile_cfg=cfg).cpu().numpy()
            assert np.array_equal(C, ref)
        # a zero segment is skipped but still consumes its rows of W
        ref0 = oracle.gemm_chain(np.ascontiguousarray(A0[np.arange(M) % mod]), W[:k0]); oracle.gemm_chain(A2, W[k0 + k1:], ref0)
        segs0 = [segs[0], gpu.operand(None, k=k1), segs[2]]
        assert np.array_equal(gpu.gemm(segs0, dW, None, M=M).cpu().numpy(), ref0)


@pytest.mark.parametrize("M,E,H", [(5, 3, 4), (64, 32, 64), (70, 12, 20), (96, 500, 1000)])
def test_lstm_cell_bitwise(gpu, oracle, M, E, H):
    rng = np.random.default_rng(M + H)
    W = rng.uniform(-.3, .3, (E + H, 4 * H)).astype(np.float32); b = rng.uniform(-.5, .5, 4 * H).astype(np.float32)
    x = rng.standard_normal((M, E)).astype(np.float32); c = rng.standard_normal((M, H)).astype(np.float32)
    h = rng.uniform(-1, 1, (M, H)).astype(np.float32)
    p = {"lstm1_W": W, "lstm1_b": b}
    vid = rng.integers(0, 1000, M).astype(np.int32); sid = rng.integers(0, 5, M).astype(np.int32)
    for keep, code in ((1.0, 0), (0.9, 258), (0.5, 600)):
        mask = None if keep >= 1 else oracle.dropout_mask(77, vid, sid, code, keep, H)
        rc, rh, rout, rg, _ = oracle.lstm1_step(p, x, c, h, mask, keep, want_gates=True)
        for cfg in range(-1, 18):                     # every tile of the table (12-16: the LDS-DMA ring tiles of round 6), 17 = out of range = auto
            gc, gh, gout, gg = gpu.lstm_cell_fwd(gpu.operand(_dev(x)), None, _dev(h), _dev(c), _dev(W), _dev(b), M, keep=keep,
                                                 seed=77, video_id=_dev(vid), sample_id=_dev(sid), drop_code=code,
                                                 want_gates=True, tile_cfg=cfg)
            assert np.array_equal(gc.cpu().numpy(), rc) and np.array_equal(gh.cpu().numpy(), rh), (keep, cfg)
            assert np.array_equal(gout.cpu().numpy(), rout), (keep, cfg)
            assert np.array_equal(gg.cpu().numpy(), rg)
    # decode-stage LSTM1: zero input (absent segment), state broadcast over samples
    rc, rh, _, _, _ = oracle.lstm1_step(p, None, np.tile(c[:3], (4, 1)), np.tile(h[:3], (4, 1)))
    gc, gh, _, _ = gpu.lstm_cell_fwd(gpu.operand(None, k=E), None, _dev(h[:3]), _dev(c[:3]), _dev(W), _dev(b), 12, state_rowmod=3)
    assert np.array_equal(gc.cpu().numpy(), rc) and np.array_equal(gh.cpu().numpy(), rh)


@pytest.mark.parametrize("M,H,V", [(6, 8, 37), (64, 64, 260), (100, 20, 97), (50, 1000, 12000)])
def test_vocab_pick_bitwise(gpu, oracle, M, H, V):
    rng = np.random.default_rng(V)
    o2 = rng.uniform(-1, 1, (M, H)).astype(np.float32); W = rng.uniform(-.1, .1, (H, V)).astype(np.float32)
    b = rng.uniform(-.1, .1, V).astype(np.float32)
    vid = rng.integers(0, 500, M).astype(np.int32); sid = rng.integers(-1, 4, M).astype(np.int32)
    logits = oracle.xw_plus_b(o2, W, b)
    ref = oracle.pick_tokens(logits, vid, sid, 5, 2024)
    for cfg in range(-1, 16):                          # every entry of fwd.hip kPick (7-14: LDS-DMA ring); 15 = out of range = auto
        tok, gl, _ = gpu.vocab_pick(_dev(o2), _dev(W), _dev(b), _dev(vid), _dev(sid), 5, 2024, want_logits=True, tile_cfg=cfg)
        assert np.array_equal(gl.cpu().numpy(), logits), cfg
        assert np.array_equal(tok.cpu().numpy(), ref), cfg


def test_vocab_pick_ties_lowest_index(gpu, oracle):
    M, H, V = 4, 4, 300
    o2 = np.zeros((M, H), np.float32); W = np.zeros((H, V), np.float32); b = np.zeros(V, np.float32)
    b[[40, 41, 200, 299]] = 3.0                        # exact ties across lanes, sub-tiles and tiles
    vid = np.zeros(M, np.int32); sid = -np.ones(M, np.int32)
    for cfg in range(-1, 16):                          # every entry of fwd.hip kPick (7-14: LDS-DMA ring); 15 = out of range = auto
        tok, _, _ = gpu.vocab_pick(_dev(o2), _dev(W), _dev(b), _dev(vid), _dev(sid), 0, 1, tile_cfg=cfg)
        assert tok.cpu().numpy().tolist() == [40] * M


SAMPLE_CASES = [
    dict(B=3, K=2, dims=dict(dim_image=24, n_words=97, word_dim=12, lstm_dim=20, n_video_lstm_step=3, n_caption_lstm_step=6)),
    dict(B=4, K=3, dims=dict(dim_image=128, n_words=260, word_dim=32, lstm_dim=64, n_video_lstm_step=5, n_caption_lstm_step=8)),
    dict(B=2, K=2, dims=dict(dim_image=1536, n_words=12000, word_dim=500, lstm_dim=1000, n_video_lstm_step=5, n_caption_lstm_step=20)),
]


@pytest.mark.parametrize("case", SAMPLE_CASES)
def test_sampler_token_ids_bit_exact(gpu, oracle, case):
    """build_multinomial_sampler x K + build_sampler: every sampled and greedy id equals the oracle's."""
    d = oracle.Dims(label_dim=0, **case["dims"])
    p = oracle.init_params(d, seed=3)
    rng = np.random.default_rng(9)
    for k in ("lstm1_b", "lstm2_b", "encode_image_b", "embed_word_b"):
        p[k] = rng.uniform(-.1, .1, p[k].shape).astype(np.float32)
    B, K = case["B"], case["K"]
    video = np.abs(rng.standard_normal((B, d.n_video_lstm_step, d.dim_image)) * 0.5).astype(np.float32)
    ref_s, ref_g = oracle.sample_captions(p, d, video, K, seed=2024, video_base=10)
    dims = gpu.make_dims(d.dim_image, d.n_words, d.word_dim, d.lstm_dim, d.n_video_lstm_step, d.n_caption_lstm_step)
    dp = {k: _dev(v) for k, v in p.items()}
    got_s, got_g = gpu.sample(dims, gpu.make_params(dp), _dev(video), K, seed=2024, video_base=10)
    assert np.array_equal(got_s.cpu().numpy(), ref_s)
    assert np.array_equal(got_g.cpu().numpy(), ref_g)
    assert len(np.unique(ref_s)) > 3                    # the draws are not degenerate
    # a different seed changes the samples but never the greedy caption
    s2, g2 = gpu.sample(dims, gpu.make_params(dp), _dev(video), K, seed=2025, video_base=10)
    assert np.array_equal(g2.cpu().numpy(), ref_g) and not np.array_equal(s2.cpu().numpy(), ref_s)


def test_operand_beyond_the_2gib_window_uses_the_scalar_path(gpu, oracle):
    """The vector path addresses each operand by 32-bit byte offsets (raw-buffer loads, 2 GiB window); an operand whose
    rows reach further must take the scalar path and still be exact.  Rows 2^28 floats apart: row 2 starts at byte 2^31."""
    import torch
    M, K, N, ld = 3, 64, 128, 1 << 28
    rng = np.random.default_rng(5)
    A = rng.standard_normal((M, K)).astype(np.float32); W = rng.standard_normal((K, N)).astype(np.float32)
    big = torch.empty(M * ld, dtype=torch.float32, device="cuda")
    view = big.view(M, ld)[:, :K]
    view.copy_(torch.as_tensor(A))
    ref = oracle.gemm_chain(A, W)
    C = gpu.gemm([gpu.operand(view)], _dev(W), None, M=M).cpu().numpy()
    assert np.array_equal(C, ref)
    # the same rows packed densely take the vector path: identical bits
    C2 = gpu.gemm([gpu.operand(_dev(A))], _dev(W), None, M=M).cpu().numpy()
    assert np.array_equal(C2, ref)


@pytest.mark.parametrize("M,K,N", SHAPES + [(320, 4000, 1000), (100, 64, 36)])
def test_gemm_nt_bitwise_all_tiles(gpu, oracle, M, K, N):
    """C = A @ Wt^T with the weight given as [N, K] (the layout the backward data-gradient products read): the same
    ascending-k chain as the plain form, on every tile configuration, vector and scalar paths, with segments."""
    rng = np.random.default_rng(M * 31 + N)
    A = rng.standard_normal((M, K)).astype(np.float32); Wt = rng.standard_normal((N, K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    ref = oracle.bias_add(oracle.gemm_chain(A, np.ascontiguousarray(Wt.T)), b)
    dA, dWt, db = _dev(A), _dev(Wt), _dev(b)
    for cfg in range(-1, 12):                           # (8-11: the LDS-DMA ring tiles with the W^T image)
        C = gpu.gemm_nt([gpu.operand(dA)], dWt, db, M=M, tile_cfg=cfg).cpu().numpy()
        assert np.array_equal(C, ref), f"tile cfg {cfg}"
    if K >= 8 and K % 8 == 0:                           # two K segments + a carried partial
        k0 = K // 2
        Ci = rng.standard_normal((M, N)).astype(np.float32)
        ref2 = Ci.copy(); oracle.gemm_chain(A, np.ascontiguousarray(Wt.T), ref2)
        segs = [gpu.operand(_dev(A[:, :k0])), gpu.operand(_dev(A[:, k0:]))]
        for cfg in (-1, 8, 9):
            C = gpu.gemm_nt(segs, dWt, None, M=M, cinit=_dev(Ci), tile_cfg=cfg).cpu().numpy()
            assert np.array_equal(C, ref2), cfg


def test_operands_larger_than_2gib_on_the_vector_path(gpu, oracle):
    """A 2.46 GB activation matrix (the reference's default B=256, K=8, Tc=35 makes a 2.9 GB logits gradient): the
    forward contraction addresses it per tile and the weight-gradient contraction per chunk, so both stay on the vector
    path and stay exact (rows beyond the 2 GiB mark included)."""
    import torch
    M, K, N = 300000, 2048, 32
    g = torch.Generator(device="cuda").manual_seed(3)
    A = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(K, N, device="cuda", generator=g)
    assert A.numel() * 4 > (1 << 31)
    C = gpu.gemm([gpu.operand(A)], W, None, M=M)
    Wh = W.cpu().numpy()
    for lo in (0, 131072 - 8, 262144 - 8, M - 40):                       # around 1 GiB, 2 GiB and the end
        rows = slice(lo, lo + 40)
        ref = oracle.gemm_chain(A[rows].cpu().numpy(), Wh)
        assert np.array_equal(C[rows].cpu().numpy(), ref), lo
    # weight-gradient form: C2[k, n] = sum_m A[m, k] * B[m, n] over all 300000 rows (order-free: vs float64)
    Bm = torch.randn(M, N, device="cuda", generator=g)
    out = torch.zeros(K, N, device="cuda")
    gpu.gemm_tn(A, Bm, out, accumulate=False)
    ref2 = (A.double().t() @ Bm.double()).cpu().numpy()
    got = out.cpu().numpy().astype(np.float64)
    assert np.abs(got - ref2).max() <= 2e-4 * np.abs(ref2).max()
