// attn_score.h -- the score chain of the temporal attention, shared by attn.hip (one launch per step) and attn_chain.hip (the
// persistent recurrence):  e = sum_h T[h] * w[h]  as ONE ascending-h fmaf chain (the numeric contract: bit-identical to the
// oracle's loop), run by a single lane on operands in LDS.
//
// The chain is 1000 dependent v_fmac_f32; what decides its speed is whether the operands are in registers when their link
// comes up.  Three register sets of 16 links rotate, each loaded TWO phases (32 links, ~160 cycles) before it is used, and
// sched_barriers keep the loads where they are written: left to itself hipcc interleaved the loads with the FMAs that need them
// and waited for LDS once or twice per 16 links (12-22 cycles per link, measured with s_memtime stamps; now ~6).
// Both rows must be padded with ZEROS to a multiple of 16 floats (a zero link adds +0: exact).
#pragma once
#include <hip/hip_runtime.h>

namespace s2vt {

typedef float sc_f32x4 __attribute__((ext_vector_type(4)));

#define S2VT_SC_LD(T_, W_, p_)                                                                   \
    do {                                                                                          \
        const int o_ = 16 * (p_);                                                                 \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                        \
            T_[j_] = *reinterpret_cast<const sc_f32x4*>(tr + o_ + 4 * j_);                        \
            W_[j_] = *reinterpret_cast<const sc_f32x4*>(wl + o_ + 4 * j_);                        \
        }                                                                                         \
    } while (0)
#define S2VT_SC_FM(T_, W_)                                                                       \
    do {                                                                                          \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                        \
            e = __builtin_fmaf(T_[j_][0], W_[j_][0], e);                                          \
            e = __builtin_fmaf(T_[j_][1], W_[j_][1], e);                                          \
            e = __builtin_fmaf(T_[j_][2], W_[j_][2], e);                                          \
            e = __builtin_fmaf(T_[j_][3], W_[j_][3], e);                                          \
        }                                                                                         \
    } while (0)

// tr, wl: LDS rows of nph * 16 floats (16-byte aligned), zero-padded behind the true length
__device__ __forceinline__ float score_chain(const float* tr, const float* wl, int nph)
{
    sc_f32x4 t0[4], w0[4], t1[4], w1[4], t2[4], w2[4];
    float e = 0.f;
    const int last = nph - 1;
    if (nph > 0) S2VT_SC_LD(t0, w0, 0);
    if (nph > 1) S2VT_SC_LD(t1, w1, 1);
    int p = 0;
    for (; p + 3 <= nph; p += 3) {
        S2VT_SC_LD(t2, w2, p + 2);
        __builtin_amdgcn_sched_barrier(0);
        S2VT_SC_FM(t0, w0);
        __builtin_amdgcn_sched_barrier(0);
        S2VT_SC_LD(t0, w0, (p + 3 < last ? p + 3 : last));
        __builtin_amdgcn_sched_barrier(0);
        S2VT_SC_FM(t1, w1);
        __builtin_amdgcn_sched_barrier(0);
        S2VT_SC_LD(t1, w1, (p + 4 < last ? p + 4 : last));
        __builtin_amdgcn_sched_barrier(0);
        S2VT_SC_FM(t2, w2);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (p < nph) { S2VT_SC_FM(t0, w0); ++p; }
    if (p < nph) { S2VT_SC_FM(t1, w1); }
    return e;
}

#undef S2VT_SC_LD
#undef S2VT_SC_FM

}  // namespace s2vt
