"""Thin torch-tensor wrappers over the C ABI (include/s2vt.h).  torch is plumbing here: device
memory, the current HIP stream, dtype checks.  All compute happens in libs2vt_hip.so."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import AttnParams, Dims, Operand, Params, check, lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk_f32(*ts):
    for t in ts:
        if t is not None:
            assert t.is_cuda and t.dtype == torch.float32, "expected a CUDA float32 tensor"


def make_dims(dim_image, n_words, word_dim, lstm_dim, n_video_lstm_step, n_caption_lstm_step, label_dim=0) -> Dims:
    return Dims(dim_image, n_words, word_dim, lstm_dim, n_video_lstm_step, n_caption_lstm_step, label_dim, 0)


def make_params(tensors: dict) -> Params:
    """tensors: name -> CUDA fp32 tensor (names = _lib.PARAM_FIELDS; missing ones become NULL)."""
    p = Params()
    for n in _lib.PARAM_FIELDS:
        t = tensors.get(n)
        if t is not None:
            _chk_f32(t)
            assert t.is_contiguous()
        setattr(p, n, None if t is None else t.data_ptr())
    return p


def operand(t, k=None, rowidx=None, rowmod=0, ld=None) -> Operand:
    """A K-segment of a concatenated operand.  t=None -> zero input of width k."""
    if t is None:
        return Operand(None, None, 0, int(k), 0, 0)
    _chk_f32(t)
    assert t.dim() == 2 and t.stride(1) == 1
    if rowidx is not None:
        assert rowidx.is_cuda and rowidx.dtype == torch.int32 and rowidx.is_contiguous()
    op = Operand(t.data_ptr(), None if rowidx is None else rowidx.data_ptr(), int(ld if ld is not None else t.stride(0)),
                 int(k if k is not None else t.shape[1]), int(rowmod), 0)
    op._keep = (t, rowidx)          # the struct holds raw pointers: keep the tensors alive with it
    return op


def math_eval(fn: str, x):
    _chk_f32(x)
    y = torch.empty_like(x)
    code = {"exp": 0, "log": 1, "tanh": 2, "sigmoid": 3}[fn]
    check(lib().s2vt_math_eval(code, _ptr(x), _ptr(y), x.numel(), _stream()), "s2vt_math_eval")
    return y


def gumbel_eval(seed, video, sample, step, V, device="cuda"):
    out = torch.empty(V, dtype=torch.float32, device=device)
    check(lib().s2vt_gumbel_eval(seed, video, sample, step, _ptr(out), V, _stream()), "s2vt_gumbel_eval")
    return out


def gemm(segs, W, bias=None, M=None, cinit=None, act_tanh=False, tile_cfg=-1, out=None):
    """C = act([seg0 ; seg1 ; seg2] @ W + bias), continuing the chain from cinit if given."""
    _chk_f32(W, bias, cinit)
    assert W.dim() == 2 and W.stride(1) == 1
    N = W.shape[1]
    arr = (Operand * len(segs))(*segs)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=W.device)
    check(lib().s2vt_gemm(arr, len(segs), _ptr(W), W.stride(0), _ptr(bias), _ptr(cinit),
                          0 if cinit is None else cinit.stride(0), _ptr(out), out.stride(0), M, N, int(act_tanh), tile_cfg,
                          _stream()), "s2vt_gemm")
    return out


def gemm_nt(segs, Wt, bias=None, M=None, cinit=None, act_tanh=False, tile_cfg=-1, out=None):
    """C = act([seg0 ; seg1 ; seg2] @ Wt^T + bias) with Wt [N, K] (K contiguous): the weight matrix as the backward reads it."""
    _chk_f32(Wt, bias, cinit)
    assert Wt.dim() == 2 and Wt.stride(1) == 1
    N = Wt.shape[0]
    arr = (Operand * len(segs))(*segs)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=Wt.device)
    check(lib().s2vt_gemm_nt(arr, len(segs), _ptr(Wt), Wt.stride(0), _ptr(bias), _ptr(cinit),
                             0 if cinit is None else cinit.stride(0), _ptr(out), out.stride(0), M, N, int(act_tanh), tile_cfg,
                             _stream()), "s2vt_gemm_nt")
    return out


def gemm_nt_splitk(A, Wt, splits=0, tile_cfg=-1, out=None, slabs=None):
    """C = A @ Wt^T with the reduction cut into K slabs that are summed (order-free: gradients only).  splits = 0: the
    library's choice for the shape."""
    _chk_f32(A, Wt)
    M, K = A.shape
    N = Wt.shape[0]
    assert Wt.shape[1] == K and A.stride(1) == 1 and Wt.stride(1) == 1
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    if slabs is None:
        slabs = torch.empty((max(splits, 12) + 1) * M * N, dtype=torch.float32, device=A.device)
    check(lib().s2vt_gemm_nt_splitk(_ptr(A), A.stride(0), _ptr(Wt), Wt.stride(0), _ptr(out), out.stride(0), M, N, K, int(splits),
                                    int(tile_cfg), _ptr(slabs), slabs.numel(), _stream()), "s2vt_gemm_nt_splitk")
    return out


def lstm_cell_fwd(x0, x1, h_prev, c_prev, W, b, M, state_rowmod=0, keep=1.0, seed=0, video_id=None, sample_id=None,
                  drop_code=0, want_gates=False, tile_cfg=-1):
    _chk_f32(h_prev, c_prev, W, b)
    H = W.shape[1] // 4
    dev = W.device
    c = torch.empty((M, H), dtype=torch.float32, device=dev)
    h = torch.empty_like(c)
    out = torch.empty_like(c)
    gates = torch.empty((M, 4 * H), dtype=torch.float32, device=dev) if want_gates else None
    check(lib().s2vt_lstm_cell_fwd(None if x0 is None else C.byref(x0), None if x1 is None else C.byref(x1), _ptr(h_prev),
                                   _ptr(c_prev), state_rowmod, _ptr(W), _ptr(b), _ptr(c), _ptr(h), _ptr(out), _ptr(gates),
                                   M, H, float(keep), seed, _ptr(video_id), _ptr(sample_id), drop_code, tile_cfg, _stream()),
          "s2vt_lstm_cell_fwd")
    return c, h, out, gates


def lstm_recurrence_fwd(W, kw0, b, h0, c0, T, cinit=None, cinit_steps=0, keep=1.0, seed=0, video_id=None, sample_id=None,
                        drop_code0=0, want_gates=True, want_out=False, persistent=-1, gates_in_cinit=False):
    """T steps of one BasicLSTMCell on the recurrent rows W[kw0:kw0+H] (+ the carried partials cinit [steps, M, 4H]).
    Returns (C_hist [T+1,M,H], H_hist [T+1,M,H], gates [T,M,4H] | None, out [T,M,H] | None).
    persistent: 1 = the one-launch persistent form, 0 = per-step launches, -1 = auto.  gates_in_cinit: write the gates over cinit."""
    _chk_f32(W, b, h0, c0, cinit)
    M, H = h0.shape
    dev = W.device
    Ch = torch.empty((T + 1, M, H), dtype=torch.float32, device=dev); Hh = torch.empty_like(Ch)
    Ch[0].copy_(c0); Hh[0].copy_(h0)
    if gates_in_cinit:
        assert cinit is not None and cinit.shape[0] == T
        gates = cinit
    else:
        gates = torch.empty((T, M, 4 * H), dtype=torch.float32, device=dev) if want_gates else None
    out = torch.empty((T, M, H), dtype=torch.float32, device=dev) if want_out else None
    nb = lib().s2vt_lstm_recurrence_scratch_bytes(H)
    ws = workspace(nb, dev, "chain")
    check(lib().s2vt_lstm_recurrence_fwd(_ptr(W), kw0, _ptr(b), _ptr(cinit), 0 if cinit is None else cinit.stride(0),
                                         0 if cinit is None else cinit.stride(1), cinit_steps, _ptr(Ch), _ptr(Hh), _ptr(gates), _ptr(out),
                                         M, H, T, float(keep), seed, _ptr(video_id), _ptr(sample_id), drop_code0, persistent, _ptr(ws),
                                         ws.numel(), _stream()), "s2vt_lstm_recurrence_fwd")
    return Ch, Hh, gates, out


def lstm_recurrence_bwd(W, kw0, gates, C_hist, dext=None, dext_t0=0, keep=1.0, seed=0, video_id=None, sample_id=None, drop_code0=0,
                        persistent=-1):
    """Back-propagation through lstm_recurrence_fwd's unroll: returns dZ [T, M, 4H].  gates [T, M, 4H] activated, C_hist
    [T+1, M, H], dext [T - dext_t0, M, H] = gradient w.r.t. the (dropped) outputs of steps dext_t0 .. T-1 or None.
    persistent: 1 = the one-launch form, 0 = per-step launches, -1 = auto."""
    _chk_f32(W, gates, C_hist, dext)
    T, M, H4 = gates.shape
    H = H4 // 4
    assert gates.is_contiguous() and C_hist.is_contiguous() and C_hist.shape == (T + 1, M, H)
    if dext is not None:
        assert dext.is_contiguous() and dext.shape == (T - dext_t0, M, H)
    dZ = torch.empty((T, M, 4 * H), dtype=torch.float32, device=W.device)
    nb = lib().s2vt_lstm_recurrence_bwd_scratch_bytes(M, H)
    ws = workspace(nb, W.device, "bchain")
    check(lib().s2vt_lstm_recurrence_bwd(_ptr(W), kw0, _ptr(gates), _ptr(C_hist), _ptr(dext), M * H, H, dext_t0, _ptr(dZ), M, H, T, float(keep),
                                         seed, _ptr(video_id), _ptr(sample_id), drop_code0, persistent, _ptr(ws), ws.numel(), _stream()),
          "s2vt_lstm_recurrence_bwd")
    return dZ


def chain_timeouts() -> int:
    """Timed-out grid-wide waits of the persistent recurrence so far (0 = healthy); synchronises the device."""
    torch.cuda.synchronize()
    return int(lib().s2vt_chain_timeouts())


def vocab_pick(out2, W, b, video_id, sample_id, step, seed, want_logits=False, tile_cfg=-1):
    _chk_f32(out2, W, b)
    M, H = out2.shape
    V = W.shape[1]
    packed = torch.zeros(M, dtype=torch.int64, device=out2.device)
    tok = torch.empty(M, dtype=torch.int32, device=out2.device)
    logits = torch.empty((M, V), dtype=torch.float32, device=out2.device) if want_logits else None
    check(lib().s2vt_vocab_pick(_ptr(out2), out2.stride(0), _ptr(W), _ptr(b), M, H, V, _ptr(video_id), _ptr(sample_id),
                                step, seed, _ptr(packed), _ptr(tok), _ptr(logits), tile_cfg, _stream()), "s2vt_vocab_pick")
    return tok, logits, packed


def frame_embed_fwd(dims: Dims, params: Params, video):
    _chk_f32(video)
    B = video.shape[0]
    emb = torch.empty((B * dims.n_video_lstm_step, dims.word_dim), dtype=torch.float32, device=video.device)
    check(lib().s2vt_frame_embed_fwd(C.byref(dims), C.byref(params), _ptr(video), B, _ptr(emb), _stream()),
          "s2vt_frame_embed_fwd")
    return emb


def zero_(*tensors):
    """Zero up to 8 device tensors (contiguous) with ONE library launch (s2vt_zero_regions)."""
    ts = [t for t in tensors if t is not None and t.numel()]
    for i in range(0, len(ts), 8):
        grp = ts[i:i + 8]
        for t in grp:
            assert t.is_cuda and t.is_contiguous() and (t.numel() * t.element_size()) % 4 == 0
        ptrs = (C.c_void_p * len(grp))(*[t.data_ptr() for t in grp])
        nb = (C.c_size_t * len(grp))(*[t.numel() * t.element_size() for t in grp])
        check(lib().s2vt_zero_regions(ptrs, nb, len(grp), _stream()), "s2vt_zero_regions")


_ws_cache = {}


def workspace(nbytes: int, device, tag="default"):
    """A cached, 256-byte aligned device scratch buffer (torch owns the memory)."""
    key = (tag, str(device))
    t = _ws_cache.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        assert t.data_ptr() % 256 == 0
        _ws_cache[key] = t
    return t


def sample(dims: Dims, params: Params, video, K: int, seed: int, video_base: int = 0, with_greedy: bool = True, stop_at_eos: bool = False):
    """K multinomial captions per video (+ greedy).  Returns (sampled [K*B,Tc], greedy [B,Tc]) int32.
    stop_at_eos (opt-in, not the reference's behaviour): a row leaves the loop once it has picked <eos>; its later ids are 0.  Ids up to
    and including the first <eos> -- the positions the objective's mask keeps -- are unchanged."""
    _chk_f32(video)
    assert video.is_contiguous()
    B = video.shape[0]
    g = 1 if with_greedy else 0
    L = lib()
    nbytes = L.s2vt_sample_workspace_bytes(C.byref(dims), B, K, g)
    ws = workspace(nbytes, video.device, "sample")
    ids = torch.empty(((K + g) * B, dims.n_caption_lstm_step), dtype=torch.int32, device=video.device)
    if stop_at_eos:
        check(L.s2vt_sample_ex(C.byref(dims), C.byref(params), _ptr(video), B, K, g, seed, video_base, 1, _ptr(ids), _ptr(ws),
                               ws.numel(), _stream()), "s2vt_sample_ex")
    else:
        check(L.s2vt_sample(C.byref(dims), C.byref(params), _ptr(video), B, K, g, seed, video_base, _ptr(ids), _ptr(ws),
                            ws.numel(), _stream()), "s2vt_sample")
    sample.serial += 1                                                 # the workspace is shared by every caller: a later call overwrites it
    sample.last_state = (ws, (K + g) * B, video.data_ptr(), B, sample.serial)   # for teacher_forced_fwd(sampler_state=...)
    return ids[:K * B], (ids[K * B:] if with_greedy else None)


sample.last_state = None
sample.serial = 0


def train_workspace(dims: Dims, B: int, N: int, device):
    nbytes = lib().s2vt_train_workspace_bytes(C.byref(dims), B, N)
    assert nbytes > 0, "bad dims / B / N (N must be a multiple of B)"
    return workspace(nbytes, device, "train")


def teacher_forced_fwd(dims: Dims, params: Params, video, caption, N: int, keep=1.0, seed=0, video_id=None,
                       sample_id=None, ws=None, logits=None, sampler_state=None, steps=None, live=None):
    """Teacher-forced unroll on N = rep*B sample-major rows.  Returns time-major logits [Tc*N, V].
    sampler_state = (workspace tensor, rows) of the sample() call of the same step on the same video block:
    LSTM1's trajectory is taken from it instead of being recomputed.
    steps (1..Tc): unroll only the first `steps` decode steps (every later position of the batch is masked); logits
    is then [steps*N, V].  caption stays [N, Tc].
    live: int32 device tensor of the unrolled rows (time-major index t*N + n, ascending) whose logits are wanted: logits is
    [len(live), V], the vocabulary projection of every other row is skipped."""
    _chk_f32(video)
    assert caption.is_cuda and caption.dtype == torch.int32 and caption.is_contiguous() and caption.shape[0] == N
    assert caption.shape[1] == dims.n_caption_lstm_step
    B = video.shape[0]
    steps = dims.n_caption_lstm_step if steps is None else int(steps)
    if ws is None:
        ws = train_workspace(dims, B, N, video.device)
    R = steps * N
    if live is not None:
        assert live.is_cuda and live.dtype == torch.int32 and live.is_contiguous() and live.dim() == 1
        R = live.numel()
    if logits is None:
        logits = torch.empty((R, dims.n_words), dtype=torch.float32, device=video.device)
    assert logits.shape[0] == R
    sws, srows = sampler_state if sampler_state is not None else (None, 0)
    check(lib().s2vt_teacher_forced_fwd_live(C.byref(dims), C.byref(params), _ptr(video), B, N, _ptr(caption), steps,
                                             _ptr(live), 0 if live is None else live.numel(), float(keep), seed,
                                              _ptr(video_id), _ptr(sample_id), _ptr(logits), _ptr(ws), ws.numel(), _ptr(sws),
                                              0 if sws is None else sws.numel(), srows, _stream()),
          "s2vt_teacher_forced_fwd")
    return logits, ws


def softmax_nll_fwd_bwd(logits, target, coef, smoothing=0.0):
    """In place: logits <- coef * (softmax - q).  Returns (nll [R], lp_target [R]).  smoothing: a float, or a CUDA fp32
    tensor [R] with one label-smoothing value per row."""
    _chk_f32(logits, coef)
    R, V = logits.shape
    assert target.dtype == torch.int32 and target.is_cuda and target.numel() == R and coef.numel() == R
    nll = torch.empty(R, dtype=torch.float32, device=logits.device)
    lp = torch.empty_like(nll)
    if isinstance(smoothing, torch.Tensor):
        _chk_f32(smoothing)
        assert smoothing.numel() == R and smoothing.is_contiguous()
        check(lib().s2vt_softmax_nll_fwd_bwd_rows(_ptr(logits), logits.stride(0), R, V, _ptr(target), _ptr(coef), _ptr(smoothing),
                                                  _ptr(nll), _ptr(lp), _stream()), "s2vt_softmax_nll_fwd_bwd_rows")
        return nll, lp
    check(lib().s2vt_softmax_nll_fwd_bwd(_ptr(logits), logits.stride(0), R, V, _ptr(target), _ptr(coef), float(smoothing),
                                         _ptr(nll), _ptr(lp), _stream()), "s2vt_softmax_nll_fwd_bwd")
    return nll, lp


def softmax_unshifted_argmax(logits, want_probs=False):
    """tf_s2vt.py:208-209 as written: argmax(exp(l) / sum(exp(l))) in fp32 without a max shift (ids int32 [R], probs)."""
    _chk_f32(logits)
    R, V = logits.shape
    ids = torch.empty(R, dtype=torch.int32, device=logits.device)
    probs = torch.empty((R, V), dtype=torch.float32, device=logits.device) if want_probs else None
    check(lib().s2vt_softmax_unshifted_argmax(_ptr(logits), logits.stride(0), R, V, _ptr(ids), _ptr(probs), _stream()),
          "s2vt_softmax_unshifted_argmax")
    return ids, probs


def bptt_bwd(dims: Dims, params: Params, grads: Params, video, N: int, dlogits, ws, keep=1.0, seed=0, video_id=None,
             sample_id=None, phase=0, steps=None, live=None):
    """phase 0 = the whole backward; 1 = vocab projection only; 2 = the rest (data-parallel overlap).
    steps: what the forward call (teacher_forced_fwd) was given."""
    _chk_f32(video, dlogits)
    steps = dims.n_caption_lstm_step if steps is None else int(steps)
    assert dlogits.shape[0] == (steps * N if live is None else live.numel())
    check(lib().s2vt_bptt_bwd_live(C.byref(dims), C.byref(params), C.byref(grads), _ptr(video), video.shape[0], N, _ptr(dlogits),
                                   steps, _ptr(live), 0 if live is None else live.numel(), float(keep), seed, _ptr(video_id),
                                   _ptr(sample_id), _ptr(ws), ws.numel(), phase, _stream()),
          "s2vt_bptt_bwd")


def bptt_dvideo(dims: Dims, params: Params, B: int, N: int, ws):
    """Gradient w.r.t. the frame features [B, Tv, dim_image] after bptt_bwd on `ws` (end-to-end CNN fine-tuning)."""
    import torch
    out = torch.empty(B, dims.n_video_lstm_step, dims.dim_image, device=ws.device, dtype=torch.float32)
    check(lib().s2vt_bptt_dvideo(C.byref(dims), C.byref(params), B, N, _ptr(out), _ptr(ws), ws.numel(), _stream()), "s2vt_bptt_dvideo")
    return out


def caption_mask(ids, mask_sum=None, want_mask=True, want_target=True, mask_sum_copy=None):
    """ids [N, Tc] int32 -> (mask [N, Tc] fp32 | None, target_tm [Tc*N] int32 | None, mask_sum [1] fp32): one launch.
    mask_sum_copy: a second place that receives sum(mask) (the gradient bucket's tail slot)."""
    assert ids.is_cuda and ids.dtype == torch.int32 and ids.is_contiguous()
    N, Tc = ids.shape
    mask = torch.empty((N, Tc), dtype=torch.float32, device=ids.device) if want_mask else None
    tgt = torch.empty(N * Tc, dtype=torch.int32, device=ids.device) if want_target else None
    if mask_sum is None:
        mask_sum = torch.empty(1, dtype=torch.float32, device=ids.device)
    check(lib().s2vt_caption_mask(_ptr(ids), N, Tc, _ptr(mask), _ptr(tgt), _ptr(mask_sum), _ptr(mask_sum_copy), _stream()), "s2vt_caption_mask")
    return mask, tgt, mask_sum


def pg_coef(mask, rewards, baseline, scale=1.0):
    _chk_f32(mask, rewards, baseline)
    N, Tc = mask.shape
    coef = torch.empty(N * Tc, dtype=torch.float32, device=mask.device)
    check(lib().s2vt_pg_coef(_ptr(mask), _ptr(rewards), _ptr(baseline), float(scale), N, Tc, _ptr(coef), _stream()), "s2vt_pg_coef")
    return coef


def xe_prep(mask, caption, loss_weight, n_global, q1):
    """(coef_tm [Tc*N], target_tm [Tc*N] int32, sum(mask) [1]) of the XE update in one launch (s2vt_xe_prep)."""
    _chk_f32(mask)
    N, Tc = mask.shape
    assert caption.is_cuda and caption.dtype == torch.int32 and caption.is_contiguous() and tuple(caption.shape) == (N, Tc)
    coef = torch.empty(N * Tc, dtype=torch.float32, device=mask.device)
    target = torch.empty(N * Tc, dtype=torch.int32, device=mask.device)
    msum = torch.empty(1, dtype=torch.float32, device=mask.device)
    check(lib().s2vt_xe_prep(_ptr(mask), _ptr(caption), N, Tc, float(loss_weight), float(n_global), int(bool(q1)), _ptr(coef), _ptr(target),
                             _ptr(msum), _stream()), "s2vt_xe_prep")
    return coef, target, msum


def mixed_prep(mask, gt_mask, rewards, baseline, sampled, gt_caption, lambda_loss, loss_weight, q1, smoothing, n_global_b):
    """The mixed objective's coefficients (s2vt_mixed_prep): (coef_tm [Tc*N], smooth_tm [Tc*N], caption_all [N, Tc] int32, target_tm [Tc*N] int32, sums [2])."""
    _chk_f32(mask, gt_mask, rewards, baseline)
    Ns, Tc = mask.shape
    B = gt_mask.shape[0]
    N = Ns + B
    for c_, n_ in ((sampled, Ns), (gt_caption, B)):
        assert c_.is_cuda and c_.dtype == torch.int32 and c_.is_contiguous() and tuple(c_.shape) == (n_, Tc)
    dev = mask.device
    coef = torch.empty(N * Tc, dtype=torch.float32, device=dev)
    smooth = torch.empty(N * Tc, dtype=torch.float32, device=dev)
    cap_all = torch.empty((N, Tc), dtype=torch.int32, device=dev)
    target = torch.empty(N * Tc, dtype=torch.int32, device=dev)
    sums = torch.empty(2, dtype=torch.float32, device=dev)
    check(lib().s2vt_mixed_prep(_ptr(mask), _ptr(gt_mask), _ptr(rewards), _ptr(baseline), _ptr(sampled), _ptr(gt_caption), Ns, B, Tc,
                                float(lambda_loss), float(loss_weight), int(bool(q1)), float(smoothing), float(n_global_b), _ptr(coef),
                                _ptr(smooth), _ptr(cap_all), _ptr(target), _ptr(sums), _stream()), "s2vt_mixed_prep")
    return coef, smooth, cap_all, target, sums


def mixed_loss(coef, nll, live, N, Ns):
    """[sum over the sampled rows, sum over the ground-truth rows, both] of coef * nll (s2vt_mixed_loss)."""
    _chk_f32(coef, nll)
    out = torch.empty(3, dtype=torch.float32, device=coef.device)
    check(lib().s2vt_mixed_loss(_ptr(coef), _ptr(nll), _ptr(live), coef.numel(), N, Ns, _ptr(out), _stream()), "s2vt_mixed_loss")
    return out


def step_scalars(coef, nll, msum_local, gsum_global, loss=None, gscale=None, sumsq=None):
    _chk_f32(coef, nll, msum_local, gsum_global, loss, gscale, sumsq)
    R = 0 if coef is None else coef.numel()
    check(lib().s2vt_step_scalars(_ptr(coef), _ptr(nll), R, _ptr(msum_local), _ptr(gsum_global), _ptr(loss), _ptr(gscale), _ptr(sumsq), _stream()),
          "s2vt_step_scalars")


def grad_finalize(g, theta, gscale, weight_decay, sumsq):
    _chk_f32(g, theta, gscale, sumsq)
    check(lib().s2vt_grad_finalize(_ptr(g), _ptr(theta), g.numel(), _ptr(gscale), float(weight_decay), _ptr(sumsq), _stream()),
          "s2vt_grad_finalize")


def adam_tf(theta, g, m, v, sumsq, clip_norm, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, applied_step=None):
    """applied_step: optional CUDA int32 tensor [1] that receives `step` when the update is applied (it is skipped on the
    device while a persistent-recurrence fault is pending, see chain_fault())."""
    _chk_f32(theta, g, m, v, sumsq)
    if applied_step is not None:
        assert applied_step.is_cuda and applied_step.dtype == torch.int32
    check(lib().s2vt_adam_tf_guarded(_ptr(theta), _ptr(g), _ptr(m), _ptr(v), theta.numel(), _ptr(sumsq), float(clip_norm), float(lr),
                                     int(step), beta1, beta2, eps, _ptr(applied_step), _stream()), "s2vt_adam_tf")


def chain_fault() -> bool:
    """True while a persistent-recurrence timeout is pending (a host-memory read: no synchronisation)."""
    return bool(lib().s2vt_chain_fault())


def chain_ack(disable_persistent: bool = True):
    """Synchronise the device, clear the fault; disable_persistent: per-step launches for the rest of the process."""
    check(lib().s2vt_chain_ack(1 if disable_persistent else 0), "s2vt_chain_ack")


class chain_hold:
    """Context manager: recurrences launched inside take their per-step form (s2vt_chain_hold) -- for the stretch of a step
    during which another stream's kernels (an asynchronous all-reduce) share the GPU with the library's stream."""

    def __enter__(self):
        check(lib().s2vt_chain_hold(1), "s2vt_chain_hold")
        return self

    def __exit__(self, *exc):
        check(lib().s2vt_chain_hold(0), "s2vt_chain_hold")
        return False


def prof_enable(on: bool):
    check(lib().s2vt_prof_enable(1 if on else 0), "s2vt_prof_enable")


def prof_filter(kernel_class: int = -1, tile_cfg: int = -1):
    check(lib().s2vt_prof_filter(kernel_class, tile_cfg), "s2vt_prof_filter")


def prof_collect():
    """Rows of the launch profiler (call after torch.cuda.synchronize())."""
    rows = (_lib.ProfRow * 64)()
    n = lib().s2vt_prof_collect(rows, 64)
    if n < 0:
        check(n, "s2vt_prof_collect")
    return [dict(kernel_class=r.kernel_class, tile_cfg=r.tile_cfg, launches=r.launches, total_ms=r.total_ms,
                 total_flops=r.total_flops, name=r.name.decode()) for r in rows[:n]]


def attention_fwd(hWa, P, Vt, w):
    """One attention step.  hWa [B,H], P/Vt [Tv,B,H], w [H] -> (scores [Tv,B], alpha [Tv,B], ctx [B,H])."""
    _chk_f32(hWa, P, Vt, w)
    Tv, B, H = P.shape
    sc = torch.empty((Tv, B), dtype=torch.float32, device=P.device)
    al = torch.empty_like(sc)
    ctx = torch.empty((B, H), dtype=torch.float32, device=P.device)
    check(lib().s2vt_attention_fwd(_ptr(hWa), _ptr(P), _ptr(Vt), _ptr(w), _ptr(sc), _ptr(al), _ptr(ctx), Tv, B, H, _stream()),
          "s2vt_attention_fwd")
    return sc, al, ctx


def attention_bwd(hWa, P, Vt, w, alpha, dctx, dw):
    """Returns (dhWa, dP, dVt); accumulates into dw [H]."""
    _chk_f32(hWa, P, Vt, w, alpha, dctx, dw)
    Tv, B, H = P.shape
    de = torch.empty((Tv, B), dtype=torch.float32, device=P.device)
    dh = torch.empty_like(hWa); dP = torch.empty_like(P); dV = torch.empty_like(Vt)
    check(lib().s2vt_attention_bwd(_ptr(hWa), _ptr(P), _ptr(Vt), _ptr(w), _ptr(alpha), _ptr(dctx), _ptr(de), _ptr(dh), _ptr(dP),
                                   _ptr(dV), _ptr(dw), Tv, B, H, _stream()), "s2vt_attention_bwd")
    return dh, dP, dV


# ---------------------------------------------------------------------------------------------
# the temporal-attention captioner as a whole model (csrc/attn_model.hip)
# ---------------------------------------------------------------------------------------------
def make_attn_params(tensors: dict) -> AttnParams:
    """tensors: name -> CUDA fp32 tensor for every name of _lib.ATTN_PARAM_FIELDS."""
    p = AttnParams()
    for n in _lib.ATTN_PARAM_FIELDS:
        t = tensors[n]
        _chk_f32(t)
        assert t.is_contiguous()
        setattr(p, n, t.data_ptr())
    p._keep = tensors
    return p


def attn_workspace(dims: Dims, B: int, device):
    nbytes = lib().s2vt_attn_workspace_bytes(C.byref(dims), B)
    assert nbytes > 0, "bad dims / B (n_video_lstm_step <= 64)"
    return workspace(nbytes, device, "attn")


def attn_teacher_forced_fwd(dims: Dims, params: AttnParams, video, caption, keep=1.0, seed=0, video_id=None, sample_id=None, steps=None,
                            ws=None, logits=None, want_alphas=False):
    """build_model's unroll (original_attention.py:88-143).  Returns (time-major logits [steps*B, V], alphas [steps,Tv,B] | None, ws)."""
    _chk_f32(video)
    B = video.shape[0]
    assert video.is_contiguous() and caption.is_cuda and caption.dtype == torch.int32 and caption.is_contiguous()
    assert caption.shape == (B, dims.n_caption_lstm_step)
    steps = dims.n_caption_lstm_step if steps is None else int(steps)
    if ws is None:
        ws = attn_workspace(dims, B, video.device)
    if logits is None:
        logits = torch.empty((steps * B, dims.n_words), dtype=torch.float32, device=video.device)
    al = torch.empty((steps, dims.n_video_lstm_step, B), dtype=torch.float32, device=video.device) if want_alphas else None
    check(lib().s2vt_attn_teacher_forced_fwd(C.byref(dims), C.byref(params), _ptr(video), B, _ptr(caption), steps, float(keep), seed,
                                             _ptr(video_id), _ptr(sample_id), _ptr(logits), _ptr(al), _ptr(ws), ws.numel(), _stream()),
          "s2vt_attn_teacher_forced_fwd")
    return logits, al, ws


def attn_loss_inputs(caption, mask, beta=0.0, want_reg=False):
    """caption [B,Tc] int32, mask [B,Tc] fp32 (device) -> (target_tm [Tc*B] int32, coef_tm [Tc*B], reg_tm | None, mask_sum [1]): one launch."""
    _chk_f32(mask)
    assert caption.is_cuda and caption.dtype == torch.int32 and caption.is_contiguous() and mask.is_contiguous() and mask.shape == caption.shape
    B, Tc = caption.shape
    dev = caption.device
    tgt = torch.empty(B * Tc, dtype=torch.int32, device=dev)
    coef = torch.empty(B * Tc, dtype=torch.float32, device=dev)
    reg = torch.empty(B * Tc, dtype=torch.float32, device=dev) if want_reg else None
    msum = torch.empty(1, dtype=torch.float32, device=dev)
    check(lib().s2vt_attn_loss_inputs(_ptr(caption), _ptr(mask), B, Tc, float(beta), _ptr(tgt), _ptr(coef), _ptr(reg), _ptr(msum), _stream()),
          "s2vt_attn_loss_inputs")
    return tgt, coef, reg, msum


def attn_step_scalars(dims: Dims, B: int, ws, coef, nll, reg_coef, reg_m, msum_local, gsum_global, loss, gscale, sumsq):
    _chk_f32(coef, nll, reg_coef, msum_local, gsum_global, loss, gscale, sumsq)
    check(lib().s2vt_attn_step_scalars(_ptr(coef), _ptr(nll), coef.numel(), _ptr(reg_coef), float(reg_m), _ptr(msum_local), _ptr(gsum_global),
                                       _ptr(loss), _ptr(gscale), _ptr(sumsq), C.byref(dims), B, _ptr(ws), ws.numel(), _stream()),
          "s2vt_attn_step_scalars")


def attn_bptt_bwd(dims: Dims, params: AttnParams, grads: AttnParams, video, dlogits, ws, steps=None, reg_coef=None, reg_m=0.5, keep=1.0,
                  seed=0, video_id=None, sample_id=None):
    _chk_f32(video, dlogits, reg_coef)
    B = video.shape[0]
    steps = dims.n_caption_lstm_step if steps is None else int(steps)
    assert dlogits.shape[0] == steps * B and (reg_coef is None or reg_coef.numel() == steps * B)
    check(lib().s2vt_attn_bptt_bwd(C.byref(dims), C.byref(params), C.byref(grads), _ptr(video), B, _ptr(dlogits), steps, _ptr(reg_coef),
                                   float(reg_m), float(keep), seed, _ptr(video_id), _ptr(sample_id), _ptr(ws), ws.numel(), _stream()),
          "s2vt_attn_bptt_bwd")


def attn_decode_greedy(dims: Dims, params: AttnParams, video, video_base=0, want_alphas=False, ws=None):
    """build_generator / build_sampler (original_attention.py:155-251): (ids [B,Tc] int32, alphas [Tc,Tv,B] | None)."""
    _chk_f32(video)
    assert video.is_contiguous()
    B = video.shape[0]
    if ws is None:
        ws = attn_workspace(dims, B, video.device)
    ids = torch.empty((B, dims.n_caption_lstm_step), dtype=torch.int32, device=video.device)
    al = torch.empty((dims.n_caption_lstm_step, dims.n_video_lstm_step, B), dtype=torch.float32, device=video.device) if want_alphas else None
    check(lib().s2vt_attn_decode_greedy(C.byref(dims), C.byref(params), _ptr(video), B, int(video_base), _ptr(ids), _ptr(al), _ptr(ws),
                                        ws.numel(), _stream()), "s2vt_attn_decode_greedy")
    return ids, al


def attr_head_fwd(video, attr_W, attr_b, labels=None):
    _chk_f32(video, attr_W, attr_b, labels)
    B, Tv, D = video.shape
    A = attr_W.shape[1]
    mean = torch.empty((B, D), dtype=torch.float32, device=video.device)
    z = torch.empty((B, A), dtype=torch.float32, device=video.device)
    bce = torch.empty_like(z) if labels is not None else None
    check(lib().s2vt_attr_head_fwd(_ptr(video), B, Tv, D, _ptr(attr_W), _ptr(attr_b), A, _ptr(labels), _ptr(mean), _ptr(z),
                                   _ptr(bce), _stream()), "s2vt_attr_head_fwd")
    return mean, z, bce


def attr_head_scores(video, attr_W, attr_b):
    """evaluate_multilabel (reinforce_multitask_e2e_attribute_loss.py:606-626): (z, sigmoid(z)) [B, A]."""
    _chk_f32(video, attr_W, attr_b)
    B, Tv, D = video.shape
    A = attr_W.shape[1]
    mean = torch.empty((B, D), dtype=torch.float32, device=video.device)
    z = torch.empty((B, A), dtype=torch.float32, device=video.device)
    scores = torch.empty_like(z)
    check(lib().s2vt_attr_head_scores(_ptr(video), B, Tv, D, _ptr(attr_W), _ptr(attr_b), A, _ptr(mean), _ptr(z), _ptr(scores), _stream()),
          "s2vt_attr_head_scores")
    return z, scores


def attr_head_bwd(mean, z, labels, scale, d_attr_W, d_attr_b):
    _chk_f32(mean, z, labels, d_attr_W, d_attr_b)
    B, D = mean.shape
    A = z.shape[1]
    dz = torch.empty_like(z)
    check(lib().s2vt_attr_head_bwd(_ptr(mean), _ptr(z), _ptr(labels), B, D, A, float(scale), _ptr(dz), _ptr(d_attr_W),
                                   _ptr(d_attr_b), _stream()), "s2vt_attr_head_bwd")
    return dz


# ---------------------------------------------------------------------------------------------
# session API + single-op entry points (include/s2vt.h, last section)
# ---------------------------------------------------------------------------------------------
class Session:
    """s2vt_create / s2vt_destroy: a handle that owns its sampler workspace.  encode once, then any
    number of greedy / multinomial decodes on that encode."""

    def __init__(self, dims: Dims, max_B: int, max_K: int):
        self.dims = dims
        self._h = C.c_void_p()
        check(lib().s2vt_create(C.byref(dims), max_B, max_K, C.byref(self._h)), "s2vt_create")
        self._B = 0

    def close(self):
        if self._h:
            check(lib().s2vt_destroy(self._h), "s2vt_destroy")
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode(self, params: Params, video):
        _chk_f32(video)
        assert video.is_contiguous()
        self._B = video.shape[0]
        check(lib().s2vt_encode_fwd(self._h, C.byref(params), _ptr(video), self._B, _stream()), "s2vt_encode_fwd")

    def decode_greedy(self, params: Params):
        ids = torch.empty((self._B, self.dims.n_caption_lstm_step), dtype=torch.int32, device="cuda")
        check(lib().s2vt_decode_greedy(self._h, C.byref(params), _ptr(ids), _stream()), "s2vt_decode_greedy")
        return ids

    def decode_multinomial(self, params: Params, K: int, seed: int, video_base: int = 0):
        ids = torch.empty((K * self._B, self.dims.n_caption_lstm_step), dtype=torch.int32, device="cuda")
        check(lib().s2vt_decode_multinomial(self._h, C.byref(params), K, seed, video_base, _ptr(ids), _stream()),
              "s2vt_decode_multinomial")
        return ids


def pack_weights(W_tf, in_dim: int):
    _chk_f32(W_tf)
    H = W_tf.shape[1] // 4
    Wx = torch.empty((in_dim, H, 4), dtype=torch.float32, device=W_tf.device)
    Wh = torch.empty((H, H, 4), dtype=torch.float32, device=W_tf.device)
    check(lib().s2vt_pack_weights(_ptr(W_tf), in_dim, H, _ptr(Wx), _ptr(Wh), _stream()), "s2vt_pack_weights")
    return Wx, Wh


def unpack_weights(Wx, Wh):
    _chk_f32(Wx, Wh)
    in_dim, H = Wx.shape[0], Wh.shape[0]
    W = torch.empty((in_dim + H, 4 * H), dtype=torch.float32, device=Wh.device)
    check(lib().s2vt_unpack_weights(_ptr(Wx), _ptr(Wh), in_dim, H, _ptr(W), _stream()), "s2vt_unpack_weights")
    return W


def frame_embed_bwd(dims: Dims, video, d_emb, dW, db):
    _chk_f32(video, d_emb, dW, db)
    check(lib().s2vt_frame_embed_bwd(C.byref(dims), _ptr(video), _ptr(d_emb), video.shape[0], _ptr(dW), _ptr(db), _stream()),
          "s2vt_frame_embed_bwd")


def lstm_cell_bwd(gates, c_new, c_prev, dh, dc_in=None):
    _chk_f32(gates, c_new, c_prev, dh, dc_in)
    M, H = c_new.shape
    dz = torch.empty((M, 4 * H), dtype=torch.float32, device=dh.device)
    dc_prev = torch.empty_like(c_new)
    check(lib().s2vt_lstm_cell_bwd(_ptr(gates), _ptr(c_new), _ptr(c_prev), _ptr(dh), _ptr(dc_in), _ptr(dz), _ptr(dc_prev), M, H,
                                   _stream()), "s2vt_lstm_cell_bwd")
    return dz, dc_prev


def xent_smooth_fwd_bwd(logits, target, coef, label_smoothing=0.05):
    _chk_f32(logits, coef)
    R, V = logits.shape
    nll = torch.empty(R, dtype=torch.float32, device=logits.device)
    check(lib().s2vt_xent_smooth_fwd_bwd(_ptr(logits), logits.stride(0), R, V, _ptr(target), _ptr(coef), float(label_smoothing),
                                         _ptr(nll), _stream()), "s2vt_xent_smooth_fwd_bwd")
    return nll


def pg_nll_fwd_bwd(logits, target_tm, adv, mask):
    """logits [Tc*N, V] time-major (overwritten with d/dlogits), target_tm int32 [Tc*N], adv [N], mask [N,Tc]."""
    _chk_f32(logits, adv, mask)
    N, Tc = mask.shape
    V = logits.shape[1]
    coef = torch.empty(N * Tc, dtype=torch.float32, device=logits.device)
    nll = torch.empty_like(coef)
    lp = torch.empty_like(coef)
    check(lib().s2vt_pg_nll_fwd_bwd(_ptr(logits), logits.stride(0), N, Tc, V, _ptr(target_tm), _ptr(adv), _ptr(mask), _ptr(coef),
                                    _ptr(nll), _ptr(lp), _stream()), "s2vt_pg_nll_fwd_bwd")
    return nll, lp, coef


def embed_gather(Wemb, idx):
    _chk_f32(Wemb)
    assert idx.dtype == torch.int32 and idx.is_cuda
    out = torch.empty((idx.numel(), Wemb.shape[1]), dtype=torch.float32, device=Wemb.device)
    check(lib().s2vt_embed_gather(_ptr(Wemb), Wemb.stride(0), _ptr(idx), idx.numel(), Wemb.shape[1], _ptr(out), out.stride(0),
                                  _stream()), "s2vt_embed_gather")
    return out


def global_norm_clip(g, clip_norm: float):
    _chk_f32(g)
    sumsq = torch.empty(1, dtype=torch.float32, device=g.device)
    check(lib().s2vt_global_norm_clip(_ptr(g), g.numel(), float(clip_norm), _ptr(sumsq), _stream()), "s2vt_global_norm_clip")
    return sumsq


def gemm_tn(A, B, C_, accumulate=True, rowidx=None):
    """C_[Kout,N] (+)= A[row(m)]^T @ B over the rows m (weight gradient of X @ W)."""
    _chk_f32(A, B, C_)
    Mred, N = B.shape
    Kout = C_.shape[0]
    check(lib().s2vt_gemm_tn(_ptr(A), A.stride(0), _ptr(rowidx), _ptr(B), B.stride(0), _ptr(C_), C_.stride(0), Mred, Kout, N,
                             1 if accumulate else 0, _stream()), "s2vt_gemm_tn")


def transpose(W):
    _chk_f32(W)
    R, Cc = W.shape
    out = torch.empty((Cc, R), dtype=torch.float32, device=W.device)
    check(lib().s2vt_transpose(_ptr(W), W.stride(0), _ptr(out), R, R, Cc, _stream()), "s2vt_transpose")
    return out


def colsum(X, out):
    _chk_f32(X, out)
    check(lib().s2vt_colsum(_ptr(X), X.stride(0), X.shape[0], X.shape[1], _ptr(out), _stream()), "s2vt_colsum")


def tanh_bwd(y, dy):
    _chk_f32(y, dy)
    dx = torch.empty_like(y)
    check(lib().s2vt_tanh_bwd(_ptr(y), _ptr(dy), _ptr(dx), y.numel(), _stream()), "s2vt_tanh_bwd")
    return dx


def dropout_bwd(dout, keep, seed, drop_code, video_id, sample_id):
    _chk_f32(dout)
    M, H = dout.shape
    dh = torch.empty((M, H), dtype=torch.float32, device=dout.device)
    check(lib().s2vt_dropout_bwd(_ptr(dout), dout.stride(0), _ptr(dh), M, H, float(keep), seed, drop_code, _ptr(video_id),
                                 _ptr(sample_id), _stream()), "s2vt_dropout_bwd")
    return dh
