"""GPU parity, random shapes: the LDS-DMA ring tiles (round 6) and the automatic tile choice against the CPU oracle -- BIT-EXACT.

tests/test_gpu_fwd.py walks every tile of the tables over a fixed list of shapes; here the shapes are drawn (seeded), so that
the eligibility rules of the ring tiles (16-byte pieces: K % 4, N % 4, row pitch, ragged last tiles, fewer rows than a tile,
fewer chunks than the ring is deep) are met from both sides: an ineligible shape must fall back to the register-staged tile of
the same geometry and give the same bits.  Reference statements: tf_s2vt.py:117 (xw_plus_b), :124-131 (the two cells), :148.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def _dims(rng, lo, hi, n, aligned_every=3):
    """n sizes in [lo, hi]; every `aligned_every`-th one a multiple of 4 (the vector / DMA paths), the rest anything."""
    out = []
    for i in range(n):
        v = int(rng.integers(lo, hi + 1))
        if i % aligned_every == 0:
            v = max(4, v // 4 * 4)
        out.append(v)
    return out


def test_store_tiles_random_shapes(gpu, oracle):
    rng = np.random.default_rng(20260601)
    Ms = _dims(rng, 1, 420, 14, aligned_every=5); Ks = _dims(rng, 1, 700, 14, aligned_every=2); Ns = _dims(rng, 1, 520, 14, aligned_every=2)
    for M, K, N in zip(Ms, Ks, Ns):
        A = rng.standard_normal((M, K)).astype(np.float32); W = rng.standard_normal((K, N)).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32)
        ref = oracle.bias_add(oracle.gemm_chain(A, W), b)
        dA, dW, db = _dev(A), _dev(W), _dev(b)
        for cfg in (-1, 8, 9, 10, 11):
            C = gpu.gemm([gpu.operand(dA)], dW, db, M=M, tile_cfg=cfg).cpu().numpy()
            assert np.array_equal(C, ref), (M, K, N, cfg)
        Wt = np.ascontiguousarray(W.T)
        dWt = _dev(Wt)
        for cfg in (-1, 8, 9, 10, 11):
            C = gpu.gemm_nt([gpu.operand(dA)], dWt, db, M=M, tile_cfg=cfg).cpu().numpy()
            assert np.array_equal(C, ref), ("nt", M, K, N, cfg)


def test_store_tiles_random_segments(gpu, oracle):
    """Three K segments (broadcast rows, gathered rows, plain rows) of random widths and a carried partial chain."""
    rng = np.random.default_rng(77001)
    for _ in range(8):
        M = int(rng.integers(1, 300)); N = int(rng.integers(1, 300)); mod = int(rng.integers(1, M + 1)); T = int(rng.integers(1, 60))
        k0, k1, k2 = (int(rng.integers(1, 40)) * int(rng.choice([1, 4])) for _ in range(3))
        A0 = rng.standard_normal((mod, k0)).astype(np.float32)
        Tab = rng.standard_normal((T, k1)).astype(np.float32); idx = rng.integers(0, T, M).astype(np.int32)
        A2 = rng.standard_normal((M, k2)).astype(np.float32)
        W = rng.standard_normal((k0 + k1 + k2, N)).astype(np.float32)
        Ci = rng.standard_normal((M, N)).astype(np.float32)
        ref = Ci.copy()
        oracle.gemm_chain(np.ascontiguousarray(A0[np.arange(M) % mod]), W[:k0], ref)
        oracle.gemm_chain(Tab, W[k0:k0 + k1], ref, rowidx=idx)
        oracle.gemm_chain(A2, W[k0 + k1:], ref)
        segs = [gpu.operand(_dev(A0), rowmod=mod), gpu.operand(_dev(Tab), rowidx=_dev(idx)), gpu.operand(_dev(A2))]
        dW, dCi = _dev(W), _dev(Ci)
        for cfg in (-1, 8, 9, 10, 11):
            C = gpu.gemm(segs, dW, None, M=M, cinit=dCi, tile_cfg=cfg).cpu().numpy()
            assert np.array_equal(C, ref), (M, N, (k0, k1, k2), mod, cfg)


def test_lstm_ring_tiles_random_shapes(gpu, oracle):
    rng = np.random.default_rng(4242)
    Ms = _dims(rng, 1, 400, 8, aligned_every=4); Es = _dims(rng, 1, 130, 8, aligned_every=2); Hs = _dims(rng, 1, 80, 8, aligned_every=1)
    for M, E, H in zip(Ms, Es, Hs):
        W = rng.uniform(-.3, .3, (E + H, 4 * H)).astype(np.float32); b = rng.uniform(-.5, .5, 4 * H).astype(np.float32)
        x = rng.standard_normal((M, E)).astype(np.float32); c = rng.standard_normal((M, H)).astype(np.float32)
        h = rng.uniform(-1, 1, (M, H)).astype(np.float32)
        vid = rng.integers(0, 1000, M).astype(np.int32); sid = rng.integers(0, 5, M).astype(np.int32)
        mask = oracle.dropout_mask(9, vid, sid, 300, 0.9, H)
        rc, rh, rout, rg, _ = oracle.lstm1_step({"lstm1_W": W, "lstm1_b": b}, x, c, h, mask, 0.9, want_gates=True)
        for cfg in (-1, 12, 13, 14, 15, 16, 17, 18):
            gc, gh, gout, gg = gpu.lstm_cell_fwd(gpu.operand(_dev(x)), None, _dev(h), _dev(c), _dev(W), _dev(b), M, keep=0.9, seed=9,
                                                 video_id=_dev(vid), sample_id=_dev(sid), drop_code=300, want_gates=True, tile_cfg=cfg)
            ok = (np.array_equal(gc.cpu().numpy(), rc) and np.array_equal(gh.cpu().numpy(), rh)
                  and np.array_equal(gout.cpu().numpy(), rout) and np.array_equal(gg.cpu().numpy(), rg))
            assert ok, (M, E, H, cfg)


def test_pick_ring_tiles_random_shapes(gpu, oracle):
    rng = np.random.default_rng(31337)
    Ms = _dims(rng, 1, 400, 8, aligned_every=4); Hs = _dims(rng, 1, 200, 8, aligned_every=2); Vs = _dims(rng, 2, 900, 8, aligned_every=2)
    for M, H, V in zip(Ms, Hs, Vs):
        o2 = rng.uniform(-1, 1, (M, H)).astype(np.float32); W = rng.uniform(-.1, .1, (H, V)).astype(np.float32)
        b = rng.uniform(-.1, .1, V).astype(np.float32)
        vid = rng.integers(0, 500, M).astype(np.int32); sid = rng.integers(-1, 4, M).astype(np.int32)
        logits = oracle.xw_plus_b(o2, W, b)
        ref = oracle.pick_tokens(logits, vid, sid, 3, 99)
        for cfg in (-1, 7, 8, 9, 10, 11, 12, 13, 14):
            tok, gl, _ = gpu.vocab_pick(_dev(o2), _dev(W), _dev(b), _dev(vid), _dev(sid), 3, 99, want_logits=True, tile_cfg=cfg)
            assert np.array_equal(gl.cpu().numpy(), logits), (M, H, V, cfg)
            assert np.array_equal(tok.cpu().numpy(), ref), (M, H, V, cfg)
