// gemm_tn.h -- weight-gradient contraction  C[k, n] (+)= sum_m A[row(m), k] * B[m, n]   (gfx950).
//
// dW = X^T . dZ over all unrolled time steps at once (K-dim = (Tv+Tc)*N rows): both operands are
// read in their natural row-major layout -- the reduction index m is the row index of BOTH, so the
// LDS images are straight copies ([m][k] and [m][n], rows == 16 (mod 32) floats apart: conflict-free
// ds_read_b32 fragment reads) and no transposed copy of the activations ever exists.  Order-free
// (gradients are compared to the oracle within a tolerance), so the reduction may be split over
// blockIdx.y with fp32 atomics when the output is too small to fill 256 CUs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_mfma.h"      // gload16 / wait_vmcnt / pin / s2vt_zero16: the asm-issued load ring

namespace s2vt {

struct TnKArgs {
    const float* A; const int* rowidx; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    int Mred, Kout, N;
    int mper;          // reduction rows per split (multiple of 32)
    int atomic;        // 1: atomicAdd into C (C pre-zeroed or accumulating); 0: plain store
    int accumulate;    // with atomic == 0: C += acc
};

template <int WM, int WN, int TM, int TN, bool VEC>
__global__ __launch_bounds__(64 * WM * WN) void gemm_tn_kernel(const TnKArgs g)
{
    constexpr int NT = 64 * WM * WN, BMo = WM * TM * 16, BNo = WN * TN * 16, BR = 32;
    constexpr int SAo = (BMo % 32 == 16) ? BMo : BMo + 16, SBo = (BNo % 32 == 16) ? BNo : BNo + 16;
    constexpr int A4 = (BR * (BMo / 4) + NT - 1) / NT, B4 = (BR * (BNo / 4) + NT - 1) / NT;
    constexpr int LPC = A4 + B4;                       // asm-issued loads per chunk per thread
    constexpr int PF = 2;                              // chunks in flight (ring slots), as the forward kernel's big tiles
    constexpr int WAITN = (PF - 1) * LPC;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                  // [2][BR][SAo]
    float* Bs = smem + 2 * BR * SAo;   // [2][BR][SBo]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, l15 = lane & 15, lq = lane >> 4;
    const int ntn = (g.N + BNo - 1) / BNo;
    const int k0 = (blockIdx.x / ntn) * BMo, n0 = (blockIdx.x % ntn) * BNo;
    const int mbeg = blockIdx.y * g.mper;
    const int mend = (mbeg + g.mper < g.Mred) ? mbeg + g.mper : g.Mred;
    const int nchunks = mend > mbeg ? (mend - mbeg + BR - 1) / BR : 0;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Staging ring: PF chunks between global memory and the LDS double buffer.  VEC: loads are inline asm (hipcc
    // would sink ordinary loads next to their LDS store and drain them with vmcnt(0)), waited for with a
    // hand-counted vmcnt; out-of-range lanes (rows beyond the split, columns beyond the matrix) read s2vt_zero16.
    f32x4 ra[PF][A4], rb[PF][B4];
    auto issue = [&](int c, f32x4 (&qa)[A4], f32x4 (&qb)[B4]) __attribute__((always_inline)) {
        const int m0 = mbeg + c * BR;
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            const int idx = tid + i * NT;
            const int r = idx / (BMo / 4), cc = (idx % (BMo / 4)) * 4;
            const int m = m0 + r, k = k0 + cc;
            const bool inr = (A4 * NT == BR * (BMo / 4) || idx < BR * (BMo / 4)) && m < mend;
            if constexpr (VEC) {
                const bool ok = inr && k < g.Kout;
                const int src = ok ? (g.rowidx ? g.rowidx[m] : m) : 0;
                gload16(qa[i], ok ? g.A + (size_t)src * g.lda + k : s2vt_zero16);
            } else {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (inr) {
                    const int src = g.rowidx ? g.rowidx[m] : m;
                    const float* p = g.A + (size_t)src * g.lda + k;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e < g.Kout) v[e] = p[e];
                }
                qa[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < B4; ++i) {
            const int idx = tid + i * NT;
            const int r = idx / (BNo / 4), cc = (idx % (BNo / 4)) * 4;
            const int m = m0 + r, n = n0 + cc;
            const bool inr = (B4 * NT == BR * (BNo / 4) || idx < BR * (BNo / 4)) && m < mend;
            if constexpr (VEC) {
                const bool ok = inr && n < g.N;
                gload16(qb[i], ok ? g.B + (size_t)m * g.ldb + n : s2vt_zero16);
            } else {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (inr) {
                    const float* p = g.B + (size_t)m * g.ldb + n;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < g.N) v[e] = p[e];
                }
                qb[i] = v;
            }
        }
    };
    auto land = [&](int buf, f32x4 (&qa)[A4], f32x4 (&qb)[B4]) __attribute__((always_inline)) {
        float* a = As + buf * BR * SAo;
        float* b = Bs + buf * BR * SBo;
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            if constexpr (VEC) pin(qa[i]);
            const int idx = tid + i * NT;
            if (A4 * NT == BR * (BMo / 4) || idx < BR * (BMo / 4))
                *reinterpret_cast<f32x4*>(a + (idx / (BMo / 4)) * SAo + (idx % (BMo / 4)) * 4) = qa[i];
        }
#pragma unroll
        for (int i = 0; i < B4; ++i) {
            if constexpr (VEC) pin(qb[i]);
            const int idx = tid + i * NT;
            if (B4 * NT == BR * (BNo / 4) || idx < BR * (BNo / 4))
                *reinterpret_cast<f32x4*>(b + (idx / (BNo / 4)) * SBo + (idx % (BNo / 4)) * 4) = qb[i];
        }
    };
    auto compute = [&](int buf, int ms0, int ms1) __attribute__((always_inline)) {
        const float* a = As + buf * BR * SAo + lq * SAo + (wm * TM) * 16 + l15;
        const float* b = Bs + buf * BR * SBo + lq * SBo + (wn * TN) * 16 + l15;
#pragma unroll
        for (int ms = ms0; ms < ms1; ++ms) {
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = a[ms * 4 * SAo + i * 16];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = b[ms * 4 * SBo + j * 16];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };

    if (nchunks > 0) {
        // prologue: chunk 0 -> LDS[0]; chunk 1 in flight in slot 1.  (Chunks beyond the split load zeros.)
        issue(0, ra[0], rb[0]);
        if constexpr (VEC) wait_vmcnt<0>();
        land(0, ra[0], rb[0]);
        issue(1, ra[1], rb[1]);
        __syncthreads();
        int c = 0;
        bool more = true;
        while (more) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                if (more) {
                    issue(c + PF, ra[j], rb[j]);                  // the slot chunk c came from
                    compute(c & 1, 0, BR / 8);
                    if constexpr (VEC) wait_vmcnt<WAITN>();       // all but the youngest chunk have returned
                    land((c + 1) & 1, ra[(j + 1) % PF], rb[(j + 1) % PF]);
                    compute(c & 1, BR / 8, BR / 4);
                    __syncthreads();
                    ++c;
                    more = c < nchunks;
                }
            }
        }
        // loads still in flight belong to chunks beyond the split: wait, and keep their registers alive until then
        if constexpr (VEC) {
            wait_vmcnt<0>();
#pragma unroll
            for (int j = 0; j < PF; ++j) {
#pragma unroll
                for (int i = 0; i < A4; ++i) pin(ra[j][i]);
#pragma unroll
                for (int i = 0; i < B4; ++i) pin(rb[j][i]);
            }
        }
    }

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 16 + l15;
        if (n >= g.N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = k0 + (wm * TM + i) * 16 + lq * 4 + r;
                if (k >= g.Kout) continue;
                float* p = g.C + (size_t)k * g.ldc + n;
                if (g.atomic) atomicAdd(p, acc[i][j][r]);
                else if (g.accumulate) *p = *p + acc[i][j][r];
                else *p = acc[i][j][r];
            }
    }
}

}  // namespace s2vt
