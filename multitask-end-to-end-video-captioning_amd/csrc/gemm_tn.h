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

namespace s2vt {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

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
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                  // [2][BR][SAo]
    float* Bs = smem + 2 * BR * SAo;   // [2][BR][SBo]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, l15 = lane & 15, lq = lane >> 4;
    const int ntn = (g.N + BNo - 1) / BNo;
    const int k0 = (blockIdx.x / ntn) * BMo, n0 = (blockIdx.x % ntn) * BNo;
    const int mbeg = blockIdx.y * g.mper;
    const int mend = (mbeg + g.mper < g.Mred) ? mbeg + g.mper : g.Mred;

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    float4 ra[A4], rb[B4];
    auto load = [&](int m0) {
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            const int idx = tid + i * NT;
            const int r = idx / (BMo / 4), c = (idx % (BMo / 4)) * 4;
            const int m = m0 + r, k = k0 + c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < BR * (BMo / 4) && m < mend) {
                const int src = g.rowidx ? g.rowidx[m] : m;
                const float* p = g.A + (size_t)src * g.lda + k;
                if (VEC) {
                    if (k < g.Kout) v = *reinterpret_cast<const float4*>(p);
                } else {
                    if (k + 0 < g.Kout) v.x = p[0];
                    if (k + 1 < g.Kout) v.y = p[1];
                    if (k + 2 < g.Kout) v.z = p[2];
                    if (k + 3 < g.Kout) v.w = p[3];
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B4; ++i) {
            const int idx = tid + i * NT;
            const int r = idx / (BNo / 4), c = (idx % (BNo / 4)) * 4;
            const int m = m0 + r, n = n0 + c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < BR * (BNo / 4) && m < mend) {
                const float* p = g.B + (size_t)m * g.ldb + n;
                if (VEC) {
                    if (n < g.N) v = *reinterpret_cast<const float4*>(p);
                } else {
                    if (n + 0 < g.N) v.x = p[0];
                    if (n + 1 < g.N) v.y = p[1];
                    if (n + 2 < g.N) v.z = p[2];
                    if (n + 3 < g.N) v.w = p[3];
                }
            }
            rb[i] = v;
        }
    };
    auto store = [&](int buf) {
        float* a = As + buf * BR * SAo;
        float* b = Bs + buf * BR * SBo;
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            const int idx = tid + i * NT;
            if (idx < BR * (BMo / 4))
                *reinterpret_cast<float4*>(a + (idx / (BMo / 4)) * SAo + (idx % (BMo / 4)) * 4) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B4; ++i) {
            const int idx = tid + i * NT;
            if (idx < BR * (BNo / 4))
                *reinterpret_cast<float4*>(b + (idx / (BNo / 4)) * SBo + (idx % (BNo / 4)) * 4) = rb[i];
        }
    };

    if (mbeg < mend) { load(mbeg); store(0); }
    __syncthreads();
    int buf = 0;
    for (int m0 = mbeg; m0 < mend; m0 += BR) {
        const bool more = m0 + BR < mend;
        if (more) load(m0 + BR);
        const float* a = As + buf * BR * SAo + lq * SAo + (wm * TM) * 16 + l15;
        const float* b = Bs + buf * BR * SBo + lq * SBo + (wn * TN) * 16 + l15;
#pragma unroll
        for (int ms = 0; ms < BR / 4; ++ms) {
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
        if (more) store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
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
