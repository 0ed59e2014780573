"""The reference's OWN default dimensions (tf_s2vt.py:304-321, reinforcement_multisampling_tf_s2vt.py:505-517,
743-753): |V| = 9,972 (msvd_vocabulary1.txt + <bos>/<eos>; 9,972 = 4 x 2,493 keeps the vector path, 9,971 would
not), n_caption_lstm_step = 35, K = 8 samples -- sampler ids and teacher-forced logits vs the oracle -- and the edge
cases of the boundary: a single video (build_generator), ragged sizes that force the scalar-load kernels, zero-size
calls, bad arguments."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def test_reference_defaults_sampler_and_logits(gpu, oracle):
    import torch
    D, V, E, H, Tv, Tc, B, K = 1536, 9972, 500, 1000, 5, 35, 3, 8
    d = oracle.Dims(D, V, E, H, Tv, Tc, 0)
    p = oracle.init_params(d, seed=21)
    rng = np.random.default_rng(3)
    p["embed_word_b"] = rng.uniform(-.5, .5, V).astype(np.float32)
    video = np.abs(rng.standard_normal((B, Tv, D)) * 0.5).astype(np.float32)
    ref_s, ref_g = oracle.sample_captions(p, d, video, K=K, seed=5)
    dims = gpu.make_dims(D, V, E, H, Tv, Tc)
    dp = {k: _dev(v) for k, v in p.items()}
    s, g = gpu.sample(dims, gpu.make_params(dp), _dev(video), K, seed=5)
    assert np.array_equal(s.cpu().numpy(), ref_s) and np.array_equal(g.cpu().numpy(), ref_g)
    cap = ref_s[:B * 2].astype(np.int32)                                   # teacher-force the first two sample blocks
    vid = np.tile(np.arange(B, dtype=np.int32), 2); sid = np.repeat(np.arange(2, dtype=np.int32), B)
    drop = oracle.dropout_masks(77, vid, sid, 0.9, H, Tv, Tc)
    ref_l = oracle.teacher_forced(p, d, np.tile(video, (2, 1, 1)), cap, drop, 0.9)
    logits, _ = gpu.teacher_forced_fwd(dims, gpu.make_params(dp), _dev(video), _dev(cap), 2 * B, 0.9, 77, _dev(vid), _dev(sid))
    assert np.array_equal(logits.view(Tc, 2 * B, V).permute(1, 0, 2).cpu().numpy(), ref_l)


def test_odd_sizes_take_the_scalar_kernels(gpu, oracle):
    """Nothing is a multiple of 4: d = 37, E = 13, H = 21, |V| = 101, B = 1 (build_generator's shape)."""
    d = oracle.Dims(37, 101, 13, 21, 3, 6, 0)
    p = oracle.init_params(d, seed=8)
    rng = np.random.default_rng(9)
    video = np.abs(rng.standard_normal((1, 3, 37))).astype(np.float32)
    ref_s, ref_g = oracle.sample_captions(p, d, video, K=2, seed=4)
    dims = gpu.make_dims(37, 101, 13, 21, 3, 6)
    dp = {k: _dev(v) for k, v in p.items()}
    s, g = gpu.sample(dims, gpu.make_params(dp), _dev(video), 2, seed=4)
    assert np.array_equal(s.cpu().numpy(), ref_s) and np.array_equal(g.cpu().numpy(), ref_g)
    cap = ref_s.astype(np.int32)
    vid = np.zeros(2, np.int32); sid = np.arange(2, dtype=np.int32)
    ref_l = oracle.teacher_forced(p, d, np.tile(video, (2, 1, 1)), cap, None, 1.0)
    logits, _ = gpu.teacher_forced_fwd(dims, gpu.make_params(dp), _dev(video), _dev(cap), 2, 1.0, 0, _dev(vid), _dev(sid))
    assert np.array_equal(logits.view(6, 2, 101).permute(1, 0, 2).cpu().numpy(), ref_l)


def test_zero_sizes_and_bad_arguments(gpu):
    import torch
    import s2vt_amd
    from s2vt_amd import _lib
    L = s2vt_amd.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = torch.zeros(8, device="cuda")
    assert L.s2vt_math_eval(0, x.data_ptr(), x.data_ptr(), 0, st) == 0                     # n = 0: nothing to do
    assert L.s2vt_global_norm_clip(x.data_ptr(), 8, 0.0, x.data_ptr(), st) == -1           # clip_norm must be > 0
    assert L.s2vt_embed_gather(x.data_ptr(), 2, x.data_ptr(), 0, 2, x.data_ptr(), 2, st) == 0
    d = _lib.Dims(16, 11, 4, 4, 2, 3, 0, 0)
    assert L.s2vt_sample(C.byref(d), None, x.data_ptr(), 1, 1, 1, 0, 0, x.data_ptr(), x.data_ptr(), 64, st) == -1   # NULL params
    p = _lib.Params(*([x.data_ptr()] * 9 + [None, None]))
    misaligned = x.data_ptr() + 4
    assert L.s2vt_sample(C.byref(d), C.byref(p), x.data_ptr(), 1, 1, 1, 0, 0, x.data_ptr(), misaligned, 1 << 20, st) == -2
    assert L.s2vt_sample(C.byref(d), C.byref(p), x.data_ptr(), 1, 1, 1, 0, 0, x.data_ptr(), x.data_ptr(), 64, st) == -3   # workspace too small
    h = C.c_void_p()
    assert L.s2vt_create(C.byref(d), 0, 1, C.byref(h)) == -1
    assert L.s2vt_destroy(None) == 0
    assert L.s2vt_error_string(-3) == b"workspace too small"
