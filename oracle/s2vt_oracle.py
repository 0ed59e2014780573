"""CPU oracle for the S2VT REINFORCE hot path -- TEST INFRASTRUCTURE, not product code.

Composes the C primitives of ``s2vt_oracle.c`` (bit-exact numeric contract: ascending-k fp32
fmaf chains, fixed-sequence exp/log/tanh/sigmoid, Philox Gumbel-max) into the graphs the
reference builds.  Every function cites the reference lines (relative to /root/reference)
it restates.  PARITY UNPINNED by the reference: TensorFlow 1.1 is absent and the reference
has no tests; see the header of s2vt_oracle.c and DESIGN.md.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libs2vt_oracle.so")
    src = os.path.join(_HERE, "s2vt_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libs2vt_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_num_threads.restype = C.c_int
    return _LIB


def _fp(a):
    return None if a is None else a.ctypes.data_as(_f32p)


def _ip(a):
    return None if a is None else a.ctypes.data_as(_i32p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# ------------------------------------------------------------------------------------------
# primitives
# ------------------------------------------------------------------------------------------
def det_exp(x):
    x = _f32(x); y = np.empty_like(x); lib().orc_expf(_fp(x), _fp(y), C.c_int64(x.size)); return y


def det_log(x):
    x = _f32(x); y = np.empty_like(x); lib().orc_logf(_fp(x), _fp(y), C.c_int64(x.size)); return y


def det_tanh(x):
    x = _f32(x); y = np.empty_like(x); lib().orc_tanhf(_fp(x), _fp(y), C.c_int64(x.size)); return y


def det_sigmoid(x):
    x = _f32(x); y = np.empty_like(x); lib().orc_sigmoidf(_fp(x), _fp(y), C.c_int64(x.size)); return y


def gemm_chain(A, W, C_=None, rowidx=None):
    """C[m,n] = fmaf chain over ascending k of A[row(m),k]*W[k,n], continuing from C_ if given.

    A and W may be row-sliced views (row stride taken from .strides); columns must be unit stride.
    """
    assert A.dtype == np.float32 and W.dtype == np.float32
    assert A.strides[1] == 4 and W.strides[1] == 4
    K, N = W.shape
    assert A.shape[1] == K
    M = A.shape[0] if rowidx is None else len(rowidx)
    acc = 0
    if C_ is None:
        C_ = np.empty((M, N), np.float32)
    else:
        assert C_.shape == (M, N) and C_.flags.c_contiguous and C_.dtype == np.float32
        acc = 1
    ri = None if rowidx is None else _i32(rowidx)
    lib().orc_gemm_chain(_fp(A), C.c_int64(A.strides[0] // 4), _ip(ri), _fp(W), C.c_int64(W.strides[0] // 4),
                         _fp(C_), C.c_int64(N), C.c_int64(M), C.c_int64(K), C.c_int64(N), C.c_int(acc))
    return C_


def bias_add(C_, b):
    b = _f32(b)
    lib().orc_bias_add(_fp(C_), C.c_int64(C_.shape[1]), _fp(b), C.c_int64(C_.shape[0]), C.c_int64(C_.shape[1]))
    return C_


def xw_plus_b(x, W, b):
    """tf.nn.xw_plus_b."""
    return bias_add(gemm_chain(_f32(x), W), b)


def lstm_pointwise(z, c_prev, drop_mask=None, keep=1.0, want_gates=False):
    M, H4 = z.shape
    H = H4 // 4
    c_prev = _f32(c_prev)
    c = np.empty((M, H), np.float32); h = np.empty((M, H), np.float32); out = np.empty((M, H), np.float32)
    gates = np.empty((M, H4), np.float32) if want_gates else None
    dm = None if drop_mask is None else _f32(drop_mask)
    lib().orc_lstm_pointwise(_fp(z), _fp(c_prev), _fp(c), _fp(h), _fp(out), _fp(dm), C.c_float(keep), _fp(gates),
                             C.c_int64(M), C.c_int64(H))
    return c, h, out, gates


def philox4x32_10(ctr, key):
    ctr = np.asarray(ctr, np.uint32); key = np.asarray(key, np.uint32); out = np.zeros(4, np.uint32)
    lib().orc_philox4x32_10(ctr.ctypes.data_as(_u32p), key.ctypes.data_as(_u32p), out.ctypes.data_as(_u32p))
    return out


def gumbel_noise(seed, v, s, t, V):
    out = np.empty(V, np.float32)
    lib().orc_gumbel_noise(C.c_uint32(seed & 0xFFFFFFFF), C.c_uint32((seed >> 32) & 0xFFFFFFFF), C.c_uint32(v),
                           C.c_uint32(s), C.c_uint32(t), _fp(out), C.c_int64(V))
    return out


def dropout_mask(seed, video_id, sample_id, code, keep, H):
    """0/1 keep mask [M,H] of the dropout Philox stream (layer*256 + step = code)."""
    vid = _i32(video_id); sid = _i32(sample_id)
    out = np.empty((len(vid), H), np.float32)
    lib().orc_dropout_mask(C.c_uint32(seed & 0xFFFFFFFF), C.c_uint32((seed >> 32) & 0xFFFFFFFF), _ip(vid), _ip(sid),
                           C.c_uint32(code), C.c_float(keep), _fp(out), C.c_int64(len(vid)), C.c_int64(H))
    return out


def dropout_masks(seed, video_id, sample_id, keep, H, Tv, Tc):
    """All masks of one teacher-forced unroll: layer 1/2 x (Tv encode + Tc decode) steps."""
    mk = lambda layer, t0, T: np.stack([dropout_mask(seed, video_id, sample_id, layer * 256 + t0 + t, keep, H) for t in range(T)])
    return {"enc1": mk(1, 0, Tv), "enc2": mk(2, 0, Tv), "dec1": mk(1, Tv, Tc), "dec2": mk(2, Tv, Tc)}


def pick_tokens(logits, video_id, sample_id, step, seed):
    M, V = logits.shape
    tok = np.empty(M, np.int32)
    vid = _i32(video_id); sid = _i32(sample_id)
    lib().orc_pick_tokens(_fp(logits), C.c_int64(V), C.c_int64(M), C.c_int64(V), _ip(vid), _ip(sid), C.c_int32(step),
                          C.c_uint32(seed & 0xFFFFFFFF), C.c_uint32((seed >> 32) & 0xFFFFFFFF), _ip(tok))
    return tok


def softmax_unshifted_argmax(logits, want_probs=False):
    """tf_s2vt.py:208-209 as written: argmax(exp(l) / sum(exp(l))), fp32, no max shift (NaN on overflow -> index 0)."""
    logits = _f32(logits)
    M, V = logits.shape
    ids = np.empty(M, np.int32)
    probs = np.empty((M, V), np.float32) if want_probs else None
    lib().orc_softmax_unshifted_argmax(_fp(logits), C.c_int64(V), C.c_int64(M), C.c_int64(V), _ip(ids), _fp(probs))
    return (ids, probs) if want_probs else ids


def row_losses(logits, target, smoothing=0.0):
    M, V = logits.shape
    nll = np.empty(M, np.float32); lp = np.empty(M, np.float32); lse = np.empty(M, np.float32)
    tg = _i32(target)
    lib().orc_row_losses(_fp(logits), C.c_int64(V), C.c_int64(M), C.c_int64(V), _ip(tg), C.c_float(smoothing),
                         _fp(nll), _fp(lp), _fp(lse))
    return nll, lp, lse


# ------------------------------------------------------------------------------------------
# parameters (tf_s2vt.py:54-88; LSTM variables per SURVEY App. B2)
# ------------------------------------------------------------------------------------------
@dataclass
class Dims:
    dim_image: int = 1536
    n_words: int = 12000
    word_dim: int = 500
    lstm_dim: int = 1000
    n_video_lstm_step: int = 5
    n_caption_lstm_step: int = 20
    label_dim: int = 400


def init_params(d: Dims, seed: int = 1234, attr: bool = False):
    """Reference initialisers: U(-0.1,0.1) for Wemb / encode_image_W / embed_word_W / attr_W
    (tf_s2vt.py:69-84), Glorot-uniform for the BasicLSTMCell kernels (TF default), zero biases."""
    rng = np.random.default_rng(seed)
    E, H, V, D = d.word_dim, d.lstm_dim, d.n_words, d.dim_image

    def u(shape, a):
        return rng.uniform(-a, a, size=shape).astype(np.float32)

    def glorot(shape):
        return u(shape, np.sqrt(6.0 / (shape[0] + shape[1])))

    p = {
        "Wemb": u((V, E), 0.1),
        "encode_image_W": u((D, E), 0.1),
        "encode_image_b": np.zeros(E, np.float32),
        "embed_word_W": u((H, V), 0.1),
        "embed_word_b": np.zeros(V, np.float32),
        "lstm1_W": glorot((E + H, 4 * H)),
        "lstm1_b": np.zeros(4 * H, np.float32),
        "lstm2_W": glorot((H + E + H, 4 * H)),
        "lstm2_b": np.zeros(4 * H, np.float32),
    }
    if attr:
        p["attr_W"] = u((D, d.label_dim), 0.1)
        p["attr_b"] = np.zeros(d.label_dim, np.float32)
    return p


# ------------------------------------------------------------------------------------------
# S2VT graph pieces
# ------------------------------------------------------------------------------------------
def frame_embed(p, video):
    """tf_s2vt.py:97-101: image_emb = reshape(video,[B*Tv,d]) @ encode_image_W + b -> [B,Tv,E].
    The tf.layers.dropout at :103 is an identity (training=False default)."""
    B, Tv, D = video.shape
    x = xw_plus_b(_f32(video).reshape(B * Tv, D), p["encode_image_W"], p["encode_image_b"])
    return x.reshape(B, Tv, -1)


def lstm1_step(p, x, c, h, drop_mask=None, keep=1.0, want_gates=False):
    """LSTM1 call (tf_s2vt.py:119,140): z = [x ; h] @ W1 + b1.  x=None means the all-zero
    `padding` input of the decode stage (:140) -- zero rows contribute nothing to the chain."""
    W = p["lstm1_W"]
    E = W.shape[0] - h.shape[1]
    if x is None:
        z = gemm_chain(_f32(h), W[E:])
    else:
        z = gemm_chain(_f32(x), W[:E])
        gemm_chain(_f32(h), W[E:], z)
    bias_add(z, p["lstm1_b"])
    return lstm_pointwise(z, c, drop_mask, keep, want_gates) + (z,)


def lstm2_step(p, out1, emb_rows, c, h, drop_mask=None, keep=1.0, want_gates=False):
    """LSTM2 call (tf_s2vt.py:122,143): z = [out1 ; e ; h2] @ W2 + b2 with e = Wemb[emb_rows]
    (decode) or the zero padding (encode, emb_rows=None)."""
    W = p["lstm2_W"]
    H = h.shape[1]
    E = W.shape[0] - 2 * H
    z = gemm_chain(_f32(out1), W[:H])
    if emb_rows is not None:
        gemm_chain(p["Wemb"], W[H:H + E], z, rowidx=emb_rows)
    gemm_chain(_f32(h), W[H + E:], z)
    bias_add(z, p["lstm2_b"])
    return lstm_pointwise(z, c, drop_mask, keep, want_gates) + (z,)


def encode(p, image_emb, drop1=None, drop2=None, keep=1.0):
    """Encoding stage (tf_s2vt.py:113-122): zero initial states (:105-107)."""
    B, Tv, _ = image_emb.shape
    H = p["lstm1_W"].shape[1] // 4
    c1 = np.zeros((B, H), np.float32); h1 = np.zeros((B, H), np.float32)
    c2 = np.zeros((B, H), np.float32); h2 = np.zeros((B, H), np.float32)
    for t in range(Tv):
        c1, h1, o1, _, _ = lstm1_step(p, image_emb[:, t], c1, h1, None if drop1 is None else drop1[t], keep)
        c2, h2, _, _, _ = lstm2_step(p, o1, None, c2, h2, None if drop2 is None else drop2[t], keep)
    return c1, h1, c2, h2


def sample_captions(p, d: Dims, video, K: int, seed: int, video_base: int = 0, with_greedy: bool = True,
                    return_logits: bool = False):
    """K multinomial decodes (build_multinomial_sampler, reinforcement_multisampling_tf_s2vt.py:294-339,
    called K times at :743-753) + one greedy decode (build_sampler, :342-391), all from ONE
    encode (the reference re-encodes per call; states are identical because samplers have no
    dropout).  Rows are sample-major (row k*B+j = sample k of video j, :764-782); the greedy
    block is last.  Returns ids [K*B, Tc] and greedy ids [B, Tc] (int32)."""
    B = video.shape[0]
    Tc = d.n_caption_lstm_step
    c1, h1, c2, h2 = encode(p, frame_embed(p, video))
    R = K + (1 if with_greedy else 0)
    tile = lambda a: np.tile(a, (R, 1))
    c1, h1, c2, h2 = tile(c1), tile(h1), tile(c2), tile(h2)
    vid = np.tile(np.arange(B, dtype=np.int32) + video_base, R)
    sid = np.repeat(np.arange(R, dtype=np.int32), B)
    if with_greedy:
        sid[K * B:] = -1
    tok = np.ones(R * B, np.int32)  # <bos> = 1 (tf_s2vt.py:129)
    ids = np.empty((R * B, Tc), np.int32)
    all_logits = []
    for t in range(Tc):
        c1, h1, o1, _, _ = lstm1_step(p, None, c1, h1)
        c2, h2, o2, _, _ = lstm2_step(p, o1, tok, c2, h2)
        logits = xw_plus_b(o2, p["embed_word_W"], p["embed_word_b"])
        tok = pick_tokens(logits, vid, sid, t, seed)
        ids[:, t] = tok
        if return_logits:
            all_logits.append(logits)
    out = (ids[:K * B], ids[K * B:] if with_greedy else None)
    return out + (np.stack(all_logits, 1),) if return_logits else out


def teacher_forced(p, d: Dims, video, caption, drop=None, keep=1.0):
    """Teacher-forced unroll shared by build_model (tf_s2vt.py:90-153) and build_loss
    (reinforcement_multisampling_tf_s2vt.py:227-292): dropout-wrapped cells when `drop` is
    given (dict with 'enc1','enc2' [Tv,N,H] and 'dec1','dec2' [Tc,N,H] 0/1 masks), previous
    word = caption[:, t-1] (<bos>=1 at t=0).  Returns logits [N, Tc, V]."""
    N = video.shape[0]
    Tc = d.n_caption_lstm_step
    g = (lambda k: None) if drop is None else (lambda k: drop[k])
    c1, h1, c2, h2 = encode(p, frame_embed(p, video), g("enc1"), g("enc2"), keep)
    caption = _i32(caption)
    logits = np.empty((N, Tc, d.n_words), np.float32)
    for t in range(Tc):
        prev = np.ones(N, np.int32) if t == 0 else caption[:, t - 1]
        c1, h1, o1, _, _ = lstm1_step(p, None, c1, h1, None if drop is None else drop["dec1"][t], keep)
        c2, h2, o2, _, _ = lstm2_step(p, o1, prev, c2, h2, None if drop is None else drop["dec2"][t], keep)
        logits[:, t] = xw_plus_b(o2, p["embed_word_W"], p["embed_word_b"])
    return logits


def xe_loss(p, d: Dims, logits, caption, mask, smoothing=0.05, loss_weight=1.0, decay=5e-5, q1=True, decay_all=False):
    """build_model's loss (tf_s2vt.py:150-166).  q1=True reproduces TF-1.1
    tf.losses.softmax_cross_entropy returning the batch MEAN (a scalar) that is then multiplied
    by the mask column; q1=False is the conventional per-row masked CE.
    Weight decay: sum l2_loss(v) for variables whose name lacks 'bias' (Q3: LSTM `biases` are
    skipped, encode_image_b / embed_word_b are decayed).  decay_all: the multitask / e2e scripts' predicate
    `if 'bias' or 'BatchNorm' not in v.name` (reinforce_multitask_e2e_attribute_s2vt.py:222, e2e_tf_s2vt.py:199) is always true:
    every variable is decayed."""
    N, Tc, V = logits.shape
    mask = np.asarray(mask, np.float64)
    tot = 0.0
    for t in range(Tc):
        nll, _, _ = row_losses(np.ascontiguousarray(logits[:, t]), caption[:, t], smoothing)
        nll = nll.astype(np.float64)
        if q1:
            tot += loss_weight * (nll.mean() * mask[:, t]).sum()
        else:
            tot += loss_weight * (nll * mask[:, t]).sum()
    wd = 0.0
    for k, v in p.items():
        if k in ("lstm1_b", "lstm2_b") and not decay_all:
            continue
        wd += 0.5 * float((v.astype(np.float64) ** 2).sum())
    return tot / mask.sum() + decay * wd


def pg_loss(logits, caption, mask, rewards, baseline):
    """REINFORCE objective (reinforcement_multisampling_tf_s2vt.py:286-291, 643-646):
    sum_loss = - sum_{n,t} lp[n,t] * mask[n,t] * (r[n]-b[n]) / sum(mask)."""
    N, Tc, V = logits.shape
    adv = np.asarray(rewards, np.float64) - np.asarray(baseline, np.float64)
    tot = 0.0
    for t in range(Tc):
        _, lp, _ = row_losses(np.ascontiguousarray(logits[:, t]), caption[:, t], 0.0)
        tot += (lp.astype(np.float64) * mask[:, t] * adv).sum()
    return -tot / np.asarray(mask, np.float64).sum()


# ------------------------------------------------------------------------------------------
# temporal attention model (original_attention.py:55-147)
# ------------------------------------------------------------------------------------------
def init_attention_params(d: Dims, seed: int = 1234):
    rng = np.random.default_rng(seed)
    H, V, D = d.lstm_dim, d.n_words, d.dim_image
    u = lambda s: rng.uniform(-0.1, 0.1, size=s).astype(np.float32)
    a = np.sqrt(6.0 / (3 * H + 4 * H))
    return {
        "Wemb": u((V, H)), "encode_image_W": u((D, H)), "encode_image_b": np.zeros(H, np.float32),
        "embed_att_w": u((H, 1)), "embed_att_Wa": u((H, H)), "embed_att_Ua": u((H, H)),
        "embed_att_ba": np.zeros(H, np.float32), "embed_word_W": u((H, V)), "embed_word_b": np.zeros(V, np.float32),
        "embed_nn_Wp": u((3 * H, H)), "embed_nn_bp": np.zeros(H, np.float32),
        "lstm3_W": rng.uniform(-a, a, size=(3 * H, 4 * H)).astype(np.float32), "lstm3_b": np.zeros(4 * H, np.float32),
    }


def attention_step(hWa, P, Vemb, w):
    Tv, B, H = P.shape
    alpha = np.empty((Tv, B), np.float32); ctx = np.empty((B, H), np.float32)
    w = _f32(w).reshape(-1)
    lib().orc_attention_step(_fp(hWa), _fp(P), _fp(Vemb), _fp(w), _fp(alpha), _fp(ctx), C.c_int64(Tv), C.c_int64(B),
                             C.c_int64(H))
    return alpha, ctx


def attention_forward(p, d: Dims, video, caption, drop=None, keep=1.0, greedy=False):
    """original_attention.py:95-147 (teacher-forced build_model) or the greedy sampler loop
    (build_generator :155-199, build_sampler :201-251) when greedy=True.
    Returns logits [B,Tc,V], alphas [Tc,Tv,B] and (greedy) ids.

    Chain order (the numeric contract, DESIGN.md section 3 "order of availability"): a pre-activation is ONE
    ascending-k fmaf chain per output over the rows of its weight matrix, the row BLOCKS taken in the order in which
    their inputs exist within the step -- inputs known before the step (the previous word's embedding), then the
    recurrent state, then what the step computes (the context, then the cell output):
        LSTM3  concat([atten, current_embed, h])  @ W3  (:131):  rows [H:2H] (embed), [2H:3H] (h), [0:H] (atten)
        output concat([output1, atten, current_embed]) @ Wp (:134):  rows [2H:3H] (embed), [H:2H] (atten), [0:H] (output1)
    (TF evaluates each as one matmul whose internal summation order is Eigen's and unknowable here; any fixed order is an
    equally faithful restatement, and this one lets the hoisted / early blocks run ahead of the recurrence.)"""
    B, Tv, D = video.shape
    H = d.lstm_dim
    Tc = d.n_caption_lstm_step
    Vt = xw_plus_b(_f32(video).reshape(B * Tv, D), p["encode_image_W"], p["encode_image_b"])
    Vt = np.ascontiguousarray(Vt.reshape(B, Tv, H).transpose(1, 0, 2))        # [Tv,B,H]  (:98)
    P = xw_plus_b(Vt.reshape(Tv * B, H), p["embed_att_Ua"], p["embed_att_ba"]).reshape(Tv, B, H)  # (:107)
    c = np.zeros((B, H), np.float32); h_prev = np.zeros((B, H), np.float32)
    q_prev = np.zeros((B, H), np.float32)                                    # attention query = previous DROPPED output (:135)
    emb = np.zeros((B, H), np.float32)                                       # (:105)
    W3 = p["lstm3_W"]; Wp = p["embed_nn_Wp"]
    logits = np.empty((B, Tc, d.n_words), np.float32); alphas = np.empty((Tc, Tv, B), np.float32)
    ids = np.empty((B, Tc), np.int32)
    for t in range(Tc):
        hWa = gemm_chain(q_prev, p["embed_att_Wa"])
        alpha, ctx = attention_step(hWa, P, Vt, p["embed_att_w"])            # (:113-128)
        z = gemm_chain(emb, W3[H:2 * H]); gemm_chain(h_prev, W3[2 * H:], z); gemm_chain(ctx, W3[:H], z)
        bias_add(z, p["lstm3_b"])
        c, h, out, _ = lstm_pointwise(z, c, None if drop is None else drop[t], keep)  # (:131-132)
        y = gemm_chain(emb, Wp[2 * H:]); gemm_chain(ctx, Wp[H:2 * H], y); gemm_chain(out, Wp[:H], y)
        bias_add(y, p["embed_nn_bp"])
        lib().orc_tanh_inplace(_fp(y), C.c_int64(y.size))                     # (:134)
        h_prev = h                                                            # the cell state carries the clean h (state_is_tuple=False)
        q_prev = out                                                          # `h_prev = output1` (:135): the DropoutWrapper output
        logits[:, t] = xw_plus_b(y, p["embed_word_W"], p["embed_word_b"])     # (:143)
        alphas[t] = alpha
        if greedy:
            tok = pick_tokens(np.ascontiguousarray(logits[:, t]), np.zeros(B, np.int32), -np.ones(B, np.int32), t, 0)
            ids[:, t] = tok
        else:
            tok = _i32(caption[:, t])
        emb = np.ascontiguousarray(p["Wemb"][tok])                            # (:141-142)
    return logits, alphas, (ids if greedy else None)


def attention_regulariser(alphas, mask, beta=10.0, m=0.5):
    """regularizer = beta * max(0, m - reduce_sum(alphas_1, 1)) * caption_mask[:, i] with alphas_1 = alpha[:, 0:8]
    (original_attention.py:118-123, 144; m = 0.5, beta = 10 at :299-300).  alphas [Tc,Tv,B], mask [B,Tc] -> reg [B,Tc]
    (fp32, the sum over the first min(8, Tv) frames in ascending order).  Identically zero while Tv <= 8: the
    unshifted softmax sums to 1 > m."""
    Tc, Tv, B = alphas.shape
    s = np.zeros((Tc, B), np.float32)
    for f in range(min(8, Tv)):
        s = (s + alphas[:, f, :]).astype(np.float32)
    hinge = np.maximum(np.float32(0.0), np.float32(m) - s).astype(np.float32)
    return (np.float32(beta) * hinge).T.astype(np.float32) * _f32(mask)


def attention_xe_loss(logits, alphas, caption, mask, beta=10.0, m=0.5):
    """build_model's loss (original_attention.py:136-150): sum_{b,t} (ce[b,t] * mask[b,t] + regularizer[b,t]) / sum(mask),
    ce = tf.nn.softmax_cross_entropy_with_logits on one-hot labels (no label smoothing here).  float64 accumulation."""
    B, Tc, V = logits.shape
    mask = np.asarray(mask, np.float64)
    tot = 0.0
    for t in range(Tc):
        nll, _, _ = row_losses(np.ascontiguousarray(logits[:, t]), caption[:, t], 0.0)
        tot += (nll.astype(np.float64) * mask[:, t]).sum()
    tot += attention_regulariser(alphas, mask, beta, m).astype(np.float64).sum()
    return tot / mask.sum()


# ------------------------------------------------------------------------------------------
# attribute head (reinforce_multitask_e2e_attribute_loss.py:375-380, 606-626)
# ------------------------------------------------------------------------------------------
def attr_head(p, video, labels=None):
    B, Tv, D = video.shape
    video = _f32(video)
    a = np.empty((B, D), np.float32)
    lib().orc_mean_frames(_fp(video), _fp(a), C.c_int64(B), C.c_int64(Tv), C.c_int64(D))
    z = xw_plus_b(a, p["attr_W"], p["attr_b"])
    if labels is None:
        return z, None
    y = _f32(labels); bce = np.empty_like(z)
    lib().orc_sigmoid_bce(_fp(z), _fp(y), _fp(bce), C.c_int64(z.size))
    return z, bce


def attr_scores(p, video):
    """evaluate_multilabel (reinforce_multitask_e2e_attribute_loss.py:621-624): scores = sigmoid(z), z as attr_head."""
    z, _ = attr_head(p, video)
    return det_sigmoid(z)


def num_threads() -> int:
    return int(lib().orc_num_threads())
