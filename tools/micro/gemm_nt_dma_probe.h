// gemm_nt_dma_probe.h (diagnostic, NOT part of the library: measured slower than the register-staged transposed-W tiles,
// profiles/r02_nt_dma_probe.jsonl) -- data-gradient contraction  C[m, k] = sum_n A[m, n] * W[k, n]   (gfx950), LDS-DMA staged.
//
// dX = dZ . W^T with W as the forward pass stores it ([k rows][n columns]): the reduction index n is the contiguous one
// of BOTH operands.  Order-free (gradients are compared to the oracle within a tolerance), which buys two things the
// exact-chain forward kernel cannot have:
//  * the chunk (32 reduction columns of BM + BN rows) goes global -> LDS by `buffer_load_dwordx4 ... lds`: no staging
//    registers, no ds_write pass, no address arithmetic in the loop;
//  * a lane reads its fragments as ONE ds_read_b128 per 16-row tile per FOUR k-steps: lane (l15, lq) takes the 4
//    consecutive floats [16g + 4lq, +4) of its row and feeds float e to MFMA e of the group -- the four MFMAs of a group
//    then reduce the columns {16g + 4q + e : q = 0..3}, a permutation of the reduction order both operands share.
// An LDS-DMA wave-instruction writes 1 KiB lane-linearly = 8 rows of 128 B, unpadded; 16 rows read at one 16-byte slot
// would hit two bank groups 8 times each, so slot s of row r is stored at slot s ^ (r & 7) (chosen through the per-lane
// SOURCE address, cdna_hip_programming.md rule 21): two rows per bank group, the floor for 128-byte rows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_mfma.h"      // make_rsrc-era helpers: kOob, wait_vmcnt, static_for, f32x4

namespace s2vt {

struct NtKArgs {
    const float* A; int lda;      // [M, Nred]
    const float* W; int ldw;      // [Kout, Nred]
    float* C; int ldc;            // [M, Kout]
    int M, Kout, Nred;
};

template <int TM, int TN, int NB>
__global__ __launch_bounds__(256) void gemm_nt_dma_kernel(const NtKArgs g)
{
    constexpr int BM = 2 * TM * 16, BN = 2 * TN * 16, BKR = 32;
    constexpr int PA = BM / 32, PB = BN / 32;             // DMA pieces per wave per chunk: 4 waves x 8 rows each
    constexpr int NM = 8 * TM * TN;                       // MFMAs per wave per chunk
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [NB][BM][32]
    float* Ws = smem + NB * BM * BKR;       // [NB][BN][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, lq = lane >> 4;
    // XCD-aware order (workgroup i runs on XCD i % 8): the column tiles of one row panel of A in consecutive slots of one XCD
    const int ncol = (g.Kout + BN - 1) / BN, nrow = (g.M + BM - 1) / BM;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int rowt = (slot / ncol) * 8 + xcd;
    if (rowt >= nrow) return;
    const int m0 = rowt * BM, k0 = (slot % ncol) * BN;
    const int nchunks = (g.Nred + BKR - 1) / BKR;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // piece p of this wave: tile rows 32p + 8*wave + (lane >> 3); physical 16-byte slot lane & 7 of the row, which holds
    // logical slot (lane & 7) ^ (row & 7)
    const int prow = 8 * wave + (lane >> 3);
    const int lcol = (((lane & 7) ^ (prow & 7)) << 2);                   // reduction column inside the chunk (rows step by 32: row & 7 == prow & 7)
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto issue_piece = [&](int c, int buf, auto p_) __attribute__((always_inline)) {
        constexpr int P = decltype(p_)::value;
        const int col = c * BKR + lcol;                                  // a chunk past the end: every lane out of range, zeros into the idle buffer
        if constexpr (P < PA) {
            const int row = P * 32 + prow;
            const bool in = col < g.Nred && m0 + row < g.M;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A + (size_t)m0 * g.lda + c * BKR), 0, (int)kOob, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(As + (buf * BM + P * 32 + 8 * wave) * BKR), 16,
                                                     in ? ((uint32_t)row * (uint32_t)g.lda + (uint32_t)lcol) * 4u : kOob, 0, 0, 0);
        } else {
            constexpr int Q = P - PA;
            const int row = Q * 32 + prow;
            const bool in = col < g.Nred && k0 + row < g.Kout;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.W + (size_t)k0 * g.ldw + c * BKR), 0, (int)kOob, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(Ws + (buf * BN + Q * 32 + 8 * wave) * BKR), 16,
                                                     in ? ((uint32_t)row * (uint32_t)g.ldw + (uint32_t)lcol) * 4u : kOob, 0, 0, 0);
        }
    };
    auto splice = [](int p) constexpr { return ((2 * p + 1) * (NM / 2)) / (2 * (PA + PB)); };   // pieces spread over the first half

    static_for<0, NB - 1>([&](auto c_) { static_for<0, PA + PB>([&](auto p_) { issue_piece(decltype(c_)::value, decltype(c_)::value, p_); }); });
    wait_vmcnt<(NB - 2) * (PA + PB)>();
    __builtin_amdgcn_s_barrier();
    // fragment addresses: row (wm*TM + i)*16 + l15, group gq (16 columns), physical slot ((4*gq + lq) ^ (l15 & 7))
    const int sl0 = ((lq ^ (l15 & 7)) << 2), sl1 = (((4 + lq) ^ (l15 & 7)) << 2);
    int buf = 0, nbuf = NB - 1;                                          // the buffer multiplied / the one chunk c+NB-1 goes to
    for (int c = 0; c < nchunks; ++c) {
        const float* a = As + (buf * BM + wm * TM * 16 + l15) * BKR;
        const float* b = Ws + (buf * BN + wn * TN * 16 + l15) * BKR;
        f32x4 av[2][TM], bv[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) av[0][i] = *reinterpret_cast<const f32x4*>(a + i * 16 * BKR + sl0);
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[0][j] = *reinterpret_cast<const f32x4*>(b + j * 16 * BKR + sl0);
        static_for<0, NM>([&](auto n_) {
            constexpr int n = decltype(n_)::value, gq = n / (4 * TM * TN), r = n % (4 * TM * TN), e = r / (TM * TN), i = (r % (TM * TN)) / TN, j = r % TN;
            if constexpr (n == 0) {
#pragma unroll
                for (int ii = 0; ii < TM; ++ii) av[1][ii] = *reinterpret_cast<const f32x4*>(a + ii * 16 * BKR + sl1);
#pragma unroll
                for (int jj = 0; jj < TN; ++jj) bv[1][jj] = *reinterpret_cast<const f32x4*>(b + jj * 16 * BKR + sl1);
            }
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[gq][i][e], bv[gq][j][e], acc[i][j], 0, 0, 0);
            static_for<0, PA + PB>([&](auto p_) {
                constexpr int p = decltype(p_)::value;
                if constexpr (splice(p) == n) issue_piece(c + NB - 1, nbuf, p_);   // no branch in the MFMA stream
            });
        });
        wait_vmcnt<(NB - 2) * (PA + PB)>();                              // chunk c+1 has landed; the later ones stay in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        nbuf = buf;
        buf = buf + 1 == NB ? 0 : buf + 1;
    }
    wait_vmcnt<0>();                                                     // the zero chunks issued past the end

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int k = k0 + (wn * TN + j) * 16 + l15;
        if (k >= g.Kout) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + (wm * TM + i) * 16 + lq * 4 + r;
                if (m < g.M) g.C[(size_t)m * g.ldc + k] = acc[i][j][r];
            }
    }
}

}  // namespace s2vt
