"""ctypes binding of libs2vt_hip.so (include/s2vt.h).  There is NO fallback: if the library is not
built, or a call fails, this raises -- the product path never routes through a CPU implementation."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class S2VTLibraryError(RuntimeError):
    pass


class S2VTChainTimeout(S2VTLibraryError):
    """A persistent recurrence (csrc/chain.hip) was starved of CUs and gave up a grid-wide wait (S2VT_E_CHAIN_TIMEOUT):
    activations computed since are suspect, variable updates queued behind it were skipped on the device.  Recover with
    Video_Caption_Generator.recover() (synchronise, acknowledge, fall back to per-step launches) and repeat the step."""


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dim_image", "n_words", "word_dim", "lstm_dim", "n_video_lstm_step",
                                         "n_caption_lstm_step", "label_dim", "reserved")]


PARAM_FIELDS = ("Wemb", "encode_image_W", "encode_image_b", "lstm1_W", "lstm1_b", "lstm2_W", "lstm2_b", "embed_word_W",
                "embed_word_b", "attr_W", "attr_b")


class Params(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in PARAM_FIELDS]


ATTN_PARAM_FIELDS = ("Wemb", "encode_image_W", "encode_image_b", "embed_att_w", "embed_att_Wa", "embed_att_Ua", "embed_att_ba",
                     "embed_word_W", "embed_word_b", "embed_nn_Wp", "embed_nn_bp", "lstm3_W", "lstm3_b")


class AttnParams(C.Structure):
    """s2vt_attn_params: the variables of original_attention.py:55-86 as device pointers."""
    _fields_ = [(n, C.c_void_p) for n in ATTN_PARAM_FIELDS]


class ProfRow(C.Structure):
    _fields_ = [("kernel_class", C.c_int32), ("tile_cfg", C.c_int32), ("launches", C.c_int64), ("total_ms", C.c_double),
                ("total_flops", C.c_double), ("name", C.c_char * 32)]


class Operand(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("rowidx", C.c_void_p), ("ld", C.c_int32), ("k", C.c_int32), ("rowmod", C.c_int32),
                ("reserved", C.c_int32)]


def lib_path() -> str:
    return os.environ.get("S2VT_LIB") or os.path.join(_HERE, "libs2vt_hip.so")   # S2VT_LIB: dev A/B builds only


_vp, _i32, _i64, _u32, _u64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float, C.c_size_t
_DP, _PP, _OP = C.POINTER(Dims), C.POINTER(Params), C.POINTER(Operand)
_AP = C.POINTER(AttnParams)

# name -> (restype, argtypes); every symbol include/s2vt.h declares
SIGNATURES = {
    "s2vt_version": (C.c_int, []),
    "s2vt_build_flags": (C.c_int, []),
    "s2vt_last_hip_error": (C.c_int, []),
    "s2vt_zero_regions": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), _i32, _vp]),
    "s2vt_error_string": (C.c_char_p, [C.c_int]),
    "s2vt_prof_enable": (C.c_int, [C.c_int]),
    "s2vt_prof_filter": (C.c_int, [C.c_int, C.c_int]),
    "s2vt_prof_collect": (C.c_int, [C.POINTER(ProfRow), C.c_int]),
    "s2vt_math_eval": (C.c_int, [C.c_int, _vp, _vp, _i64, _vp]),
    "s2vt_gumbel_eval": (C.c_int, [_u64, _i32, _i32, _i32, _vp, _i32, _vp]),
    "s2vt_gemm": (C.c_int, [_OP, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "s2vt_gemm_nt": (C.c_int, [_OP, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "s2vt_lstm_cell_fwd": (C.c_int, [_OP, _OP, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _u64, _vp,
                                     _vp, _u32, _i32, _vp]),
    "s2vt_vocab_pick": (C.c_int, [_vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _u64, _vp, _vp, _vp, _i32, _vp]),
    "s2vt_frame_embed_fwd": (C.c_int, [_DP, _PP, _vp, _i32, _vp, _vp]),
    "s2vt_sample_workspace_bytes": (_sz, [_DP, _i32, _i32, _i32]),
    "s2vt_sample": (C.c_int, [_DP, _PP, _vp, _i32, _i32, _i32, _u64, _i32, _vp, _vp, _sz, _vp]),
    "s2vt_sample_ex": (C.c_int, [_DP, _PP, _vp, _i32, _i32, _i32, _u64, _i32, _i32, _vp, _vp, _sz, _vp]),
    "s2vt_train_workspace_bytes": (_sz, [_DP, _i32, _i32]),
    "s2vt_teacher_forced_fwd": (C.c_int, [_DP, _PP, _vp, _i32, _i32, _vp, _f32, _u64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "s2vt_teacher_forced_fwd_reuse": (C.c_int, [_DP, _PP, _vp, _i32, _i32, _vp, _f32, _u64, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _i32, _vp]),
    "s2vt_gemm_nt_splitk": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "s2vt_teacher_forced_fwd_steps": (C.c_int, [_DP, _PP, _vp, _i32, _i32, _vp, _i32, _f32, _u64, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _i32, _vp]),
    "s2vt_teacher_forced_fwd_live": (C.c_int, [_DP, _PP, _vp, _i32, _i32, _vp, _i32, _vp, _i32, _f32, _u64, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _i32, _vp]),
    "s2vt_bptt_bwd_live": (C.c_int, [_DP, _PP, _PP, _vp, _i32, _i32, _vp, _i32, _vp, _i32, _f32, _u64, _vp, _vp, _vp, _sz, _i32, _vp]),
    "s2vt_bptt_bwd_steps": (C.c_int, [_DP, _PP, _PP, _vp, _i32, _i32, _vp, _i32, _f32, _u64, _vp, _vp, _vp, _sz, _i32, _vp]),
    "s2vt_softmax_nll_fwd_bwd": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _f32, _vp, _vp, _vp]),
    "s2vt_softmax_unshifted_argmax": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "s2vt_softmax_nll_fwd_bwd_rows": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "s2vt_bptt_bwd": (C.c_int, [_DP, _PP, _PP, _vp, _i32, _i32, _vp, _f32, _u64, _vp, _vp, _vp, _sz, _vp]),
    "s2vt_bptt_bwd_phase": (C.c_int, [_DP, _PP, _PP, _vp, _i32, _i32, _vp, _f32, _u64, _vp, _vp, _vp, _sz, _i32, _vp]),
    "s2vt_bptt_dvideo": (C.c_int, [_DP, _PP, _i32, _i32, _vp, _vp, _sz, _vp]),
    "s2vt_embed_scatter_add": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _vp, _vp]),
    "s2vt_caption_mask": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "s2vt_pg_coef": (C.c_int, [_vp, _vp, _vp, _f32, _i32, _i32, _vp, _vp]),
    "s2vt_xe_prep": (C.c_int, [_vp, _vp, _i32, _i32, _f32, _f32, _i32, _vp, _vp, _vp, _vp]),
    "s2vt_mixed_prep": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, C.c_double, _f32, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "s2vt_mixed_loss": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "s2vt_step_scalars": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "s2vt_grad_finalize": (C.c_int, [_vp, _vp, _i64, _vp, _f32, _vp, _vp]),
    "s2vt_adam_tf": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _f32, _f32, _i64, _f32, _f32, _f32, _vp]),
    "s2vt_attention_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "s2vt_attention_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "s2vt_attn_workspace_bytes": (_sz, [_DP, _i32]),
    "s2vt_attn_teacher_forced_fwd": (C.c_int, [_DP, _AP, _vp, _i32, _vp, _i32, _f32, _u64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "s2vt_attn_loss_inputs": (C.c_int, [_vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "s2vt_attn_step_scalars": (C.c_int, [_vp, _vp, _i64, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _DP, _i32, _vp, _sz, _vp]),
    "s2vt_attn_bptt_bwd": (C.c_int, [_DP, _AP, _AP, _vp, _i32, _vp, _i32, _vp, _f32, _f32, _u64, _vp, _vp, _vp, _sz, _vp]),
    "s2vt_attn_decode_greedy": (C.c_int, [_DP, _AP, _vp, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "s2vt_attr_head_fwd": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "s2vt_attr_head_bwd": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp]),
    "s2vt_attr_head_scores": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "s2vt_create": (C.c_int, [_DP, _i32, _i32, C.POINTER(_vp)]),
    "s2vt_destroy": (C.c_int, [_vp]),
    "s2vt_encode_fwd": (C.c_int, [_vp, _PP, _vp, _i32, _vp]),
    "s2vt_decode_greedy": (C.c_int, [_vp, _PP, _vp, _vp]),
    "s2vt_decode_multinomial": (C.c_int, [_vp, _PP, _i32, _u64, _i32, _vp, _vp]),
    "s2vt_pack_weights": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "s2vt_unpack_weights": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp]),
    "s2vt_frame_embed_bwd": (C.c_int, [_DP, _vp, _vp, _i32, _vp, _vp, _vp]),
    "s2vt_lstm_cell_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "s2vt_xent_smooth_fwd_bwd": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _f32, _vp, _vp]),
    "s2vt_pg_nll_fwd_bwd": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "s2vt_embed_gather": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _vp, _i32, _vp]),
    "s2vt_global_norm_clip": (C.c_int, [_vp, _i64, _f32, _vp, _vp]),
    "s2vt_gemm_tn": (C.c_int, [_vp, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "s2vt_transpose": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _vp]),
    "s2vt_colsum": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "s2vt_tanh_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "s2vt_dropout_bwd": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _f32, _u64, _u32, _vp, _vp, _vp]),
    "s2vt_lstm_recurrence_scratch_bytes": (_sz, [_i32]),
    "s2vt_lstm_recurrence_fwd": (C.c_int, [_vp, _i32, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _u64, _vp, _vp,
                                           _u32, _i32, _vp, _sz, _vp]),
    "s2vt_lstm_recurrence_bwd_scratch_bytes": (_sz, [_i32, _i32]),
    "s2vt_lstm_recurrence_bwd": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _i32, _i32, _i32, _f32, _u64, _vp, _vp, _u32, _i32,
                                           _vp, _sz, _vp]),
    "s2vt_chain_timeouts": (C.c_int, []),
    "s2vt_chain_fault": (C.c_int, []),
    "s2vt_chain_ack": (C.c_int, [C.c_int]),
    "s2vt_chain_hold": (C.c_int, [C.c_int]),
    "s2vt_adam_tf_guarded": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _f32, _f32, _i64, _f32, _f32, _f32, _vp, _vp]),
    "s2vt_allreduce_grads": (C.c_int, [_vp, _i64, _vp, _vp]),
    "s2vt_set_rccl_allreduce": (C.c_int, [_vp]),
}


def lib():
    """Load libs2vt_hip.so (built by __graft_entry__.build() / csrc/Makefile).  Raises if absent."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise S2VTLibraryError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        try:
            L = C.CDLL(path)
        except OSError as e:  # pragma: no cover
            raise S2VTLibraryError(f"cannot load {path}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(L, name)
            except AttributeError as e:
                raise S2VTLibraryError(f"{path} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc: int, what: str = "s2vt call"):
    if rc != 0:
        L = lib()
        msg = L.s2vt_error_string(rc).decode()
        extra = f" hipError={L.s2vt_last_hip_error()}" if rc == -4 else ""
        if rc == -5:
            raise S2VTChainTimeout(f"{what} refused: {msg} (code {rc})")
        raise S2VTLibraryError(f"{what} failed: {msg} (code {rc}){extra}")
