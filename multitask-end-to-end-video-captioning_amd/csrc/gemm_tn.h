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

#include "gemm_mfma.h"      // bload16 / make_rsrc / wait_vmcnt / pin / static_for: the asm-issued load ring

namespace s2vt {

struct TnKArgs {
    const float* A; const int* rowidx; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    int Mred, Kout, N;
    int mper;          // reduction rows per split (multiple of 32)
    int splits;        // reduction slabs
    int xmap;          // 1: one-dimensional grid, XCD-aware workgroup -> (row panel, column panel, slab) order
    int atomic;        // 1: atomicAdd into C (C pre-zeroed or accumulating); 0: plain store
    int accumulate;    // with atomic == 0: C += acc
    int prio_rot;      // > 0: the workgroup's waves rotate their issue priority every chunk (see gemm_tn_dma_kernel)
    float* colsum;     // optional [N]: += sum_m B[m, n] (the bias gradient that goes with this weight gradient: the B tiles are
                       // in registers anyway); added by the workgroups of the first row panel, fp32 atomics; vector path only
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
    int k0, n0, mbeg;
    if (g.xmap) {
        // XCD-aware order (workgroup i runs on XCD i % 8): the row panels of one (column panel, reduction slab) pair sit in
        // consecutive slots of ONE XCD, so that slab of B crosses the fabric once and the XCD's L2 serves the other panels;
        // an XCD takes a CONTIGUOUS range of pairs (slab-major), so the pairs it works on belong to one or two slabs and
        // share those slabs' A slices (dealt round-robin, every XCD touched every slab: all of A, eight times).
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int nrow = (g.Kout + BMo - 1) / BMo;
        const int npr = ntn * g.splits;
        const int pr = xcd * ((npr + 7) >> 3) + slot / nrow;
        if (slot / nrow >= ((npr + 7) >> 3) || pr >= npr) return;
        k0 = (slot % nrow) * BMo; n0 = (pr % ntn) * BNo; mbeg = (pr / ntn) * g.mper;
    } else {
        k0 = (blockIdx.x / ntn) * BMo; n0 = (blockIdx.x % ntn) * BNo; mbeg = blockIdx.y * g.mper;
    }
    const int mend = (mbeg + g.mper < g.Mred) ? mbeg + g.mper : g.Mred;
    const int nchunks = mend > mbeg ? (mend - mbeg + BR - 1) / BR : 0;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Staging ring: PF chunks between global memory and the LDS double buffer (vector path: asm-issued raw-buffer
    // loads with a hand-counted vmcnt, below; this scalar form serves unaligned / odd shapes).
    f32x4 ra[PF][A4], rb[PF][B4];
    auto issue = [&](int c, f32x4 (&qa)[A4], f32x4 (&qb)[B4]) __attribute__((always_inline)) {
        const int m0 = mbeg + c * BR;
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            const int idx = tid + i * NT;
            const int r = idx / (BMo / 4), cc = (idx % (BMo / 4)) * 4;
            const int m = m0 + r, k = k0 + cc;
            const bool inr = (A4 * NT == BR * (BMo / 4) || idx < BR * (BMo / 4)) && m < mend;
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
#pragma unroll
        for (int i = 0; i < B4; ++i) {
            const int idx = tid + i * NT;
            const int r = idx / (BNo / 4), cc = (idx % (BNo / 4)) * 4;
            const int m = m0 + r, n = n0 + cc;
            const bool inr = (B4 * NT == BR * (BNo / 4) || idx < BR * (BNo / 4)) && m < mend;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (inr) {
                const float* p = g.B + (size_t)m * g.ldb + n;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < g.N) v[e] = p[e];
            }
            qb[i] = v;
        }
    };
    auto land = [&](int buf, f32x4 (&qa)[A4], f32x4 (&qb)[B4]) __attribute__((always_inline)) {
        float* a = As + buf * BR * SAo;
        float* b = Bs + buf * BR * SBo;
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            const int idx = tid + i * NT;
            if (A4 * NT == BR * (BMo / 4) || idx < BR * (BMo / 4))
                *reinterpret_cast<f32x4*>(a + (idx / (BMo / 4)) * SAo + (idx % (BMo / 4)) * 4) = qa[i];
        }
#pragma unroll
        for (int i = 0; i < B4; ++i) {
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

    if constexpr (VEC) {
      if (nchunks > 0) {
        // ---- interleaved loop (same structure as gemm_kernel's, see there): the chunk's side work is cut into A4+B4
        // pieces spliced after fixed MFMAs -- first half the raw-buffer loads of chunk c+PF (out-of-range lanes get
        // zeros from the bounds check, the per-lane offsets are loop-invariant unless A is gathered), then the
        // counted vmcnt wait, second half the LDS stores of chunk c+1; fragments are read one k-step ahead.
        constexpr int KS = BR / 4, MPK = TM * TN, NM = KS * MPK, HALF = NM / 2;
        // descriptors are re-based on the first row of every chunk (64-bit scalar add), so the 32-bit offsets span one
        // chunk (32 rows) whatever the size of the matrices
        i32x4 rsA = make_rsrc(g.rowidx ? g.A : g.A + (size_t)mbeg * g.lda), rsB = make_rsrc(g.B + (size_t)mbeg * g.ldb);
        auto rebase = [](i32x4& r, uint32_t bytes) __attribute__((always_inline)) {
            const uint64_t b = (((uint64_t)(uint32_t)r[1] << 32) | (uint32_t)r[0]) + bytes;
            r[0] = (int)(uint32_t)b;
            r[1] = (int)(uint32_t)(b >> 32);
        };
        uint32_t avo[A4], bvo[B4];
        int ar[A4], br[B4];
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            const int idx = tid + i * NT;
            const int r = idx / (BMo / 4), k = k0 + (idx % (BMo / 4)) * 4;
            const bool ok = (A4 * NT == BR * (BMo / 4) || idx < BR * (BMo / 4)) && k < g.Kout;
            ar[i] = ok ? r : (1 << 30);                                  // a row no split ever reaches
            avo[i] = g.rowidx ? (uint32_t)k * 4u : (uint32_t)(r * g.lda + k) * 4u;
        }
#pragma unroll
        for (int i = 0; i < B4; ++i) {
            const int idx = tid + i * NT;
            const int r = idx / (BNo / 4), n = n0 + (idx % (BNo / 4)) * 4;
            const bool ok = (B4 * NT == BR * (BNo / 4) || idx < BR * (BNo / 4)) && n < g.N;
            br[i] = ok ? r : (1 << 30);
            bvo[i] = (uint32_t)(r * g.ldb + n) * 4u;
        }
        int mrem = mend - mbeg;                                          // rows left from the next chunk to issue
        int mnext = mbeg;
        const uint32_t stepA = g.rowidx ? 0u : (uint32_t)BR * (uint32_t)g.lda * 4u, stepB = (uint32_t)BR * (uint32_t)g.ldb * 4u;
        auto issue_piece = [&](auto p_, f32x4 (&qa)[A4], f32x4 (&qb)[B4]) __attribute__((always_inline)) {
            constexpr int P = decltype(p_)::value;
            if constexpr (P < A4) {
                uint32_t vo = avo[P];
                if (g.rowidx) {                                          // gathered rows (frame-embedding gradient): index load per chunk
                    const bool in = ar[P] < mrem;
                    vo += (uint32_t)(in ? g.rowidx[mnext + ar[P]] : 0) * (uint32_t)g.lda * 4u;
                }
                bload16(qa[P], ar[P] < mrem ? vo : kOob, rsA, 0u);
            } else {
                constexpr int i = P - A4;
                bload16(qb[i], br[i] < mrem ? bvo[i] : kOob, rsB, 0u);
            }
        };
        auto walk_next = [&]() __attribute__((always_inline)) {
            mrem -= BR;
            mnext += BR;
            rebase(rsA, stepA);
            rebase(rsB, stepB);
        };
        auto land_piece = [&](int buf, auto p_, f32x4 (&qa)[A4], f32x4 (&qb)[B4]) __attribute__((always_inline)) {
            constexpr int P = decltype(p_)::value;
            if constexpr (P < A4) {
                pin(qa[P]);
                const int idx = tid + P * NT;
                if (A4 * NT == BR * (BMo / 4) || idx < BR * (BMo / 4))
                    *reinterpret_cast<f32x4*>(As + buf * BR * SAo + (idx / (BMo / 4)) * SAo + (idx % (BMo / 4)) * 4) = qa[P];
            } else {
                constexpr int i = P - A4;
                pin(qb[i]);
                const int idx = tid + i * NT;
                if (B4 * NT == BR * (BNo / 4) || idx < BR * (BNo / 4))
                    *reinterpret_cast<f32x4*>(Bs + buf * BR * SBo + (idx / (BNo / 4)) * SBo + (idx % (BNo / 4)) * 4) = qb[i];
            }
        };
        auto splice = [](int p) constexpr { return ((2 * p + 1) * HALF) / (2 * LPC); };
        // column sums of B beside the MFMAs (free VALU slots): the waves of the first wave-row of the first row panel
        const float csm = (g.colsum && k0 == 0 && wm == 0) ? 1.0f : 0.0f;
        float cs[TN];
#pragma unroll
        for (int jj = 0; jj < TN; ++jj) cs[jj] = 0.f;

        static_for<0, LPC>([&](auto p_) { issue_piece(p_, ra[0], rb[0]); });
        walk_next();
        wait_vmcnt<0>();
        land(0, ra[0], rb[0]);
        static_for<0, LPC>([&](auto p_) { issue_piece(p_, ra[1], rb[1]); });
        walk_next();
        __syncthreads();
        // two copies of the main loop: only the waves that own a bias column sum (first row panel, first k slab) carry its adds
        auto main_loop = [&](auto cs_) __attribute__((always_inline)) {
        constexpr bool CS = decltype(cs_)::value;
        int c = 0;
        bool more = true;
        while (more) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                if (more) {
                    const int buf = c & 1;
                    const float* a = As + buf * BR * SAo + lq * SAo + (wm * TM) * 16 + l15;
                    const float* b = Bs + buf * BR * SBo + lq * SBo + (wn * TN) * 16 + l15;
                    float av[2][TM], bv[2][TN];
                    auto read_frag = [&](auto ks_, float (&qa)[TM], float (&qb)[TN]) __attribute__((always_inline)) {
                        constexpr int ks = decltype(ks_)::value;
#pragma unroll
                        for (int i = 0; i < TM; ++i) qa[i] = a[ks * 4 * SAo + i * 16];
#pragma unroll
                        for (int jj = 0; jj < TN; ++jj) qb[jj] = b[ks * 4 * SBo + jj * 16];
                    };
                    read_frag(std::integral_constant<int, 0>{}, av[0], bv[0]);
                    static_for<0, NM>([&](auto n_) {
                        constexpr int n = decltype(n_)::value, ks = n / MPK, r = n % MPK, i = r / TN, jj = r % TN;
                        if constexpr (r == 0 && ks + 1 < KS) {
                            read_frag(std::integral_constant<int, ks + 1>{}, av[(ks + 1) & 1], bv[(ks + 1) & 1]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if constexpr (n == HALF) wait_vmcnt<WAITN>();
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks & 1][i], bv[ks & 1][jj], acc[i][jj], 0, 0, 0);
                        if constexpr (CS && i == 0) cs[jj] += bv[ks & 1][jj];
                        static_for<0, LPC>([&](auto p_) {
                            constexpr int p = decltype(p_)::value;
                            if constexpr (splice(p) == n) issue_piece(p_, ra[j], rb[j]);
                            if constexpr (HALF + splice(p) == n) land_piece((c + 1) & 1, p_, ra[(j + 1) % PF], rb[(j + 1) % PF]);
                        });
                    });
                    walk_next();
                    __syncthreads();
                    ++c;
                    more = c < nchunks;
                }
            }
        }
        };
        if (csm != 0.0f) main_loop(std::true_type{});
        else main_loop(std::false_type{});
        wait_vmcnt<0>();
#pragma unroll
        for (int j = 0; j < PF; ++j) {
#pragma unroll
            for (int i = 0; i < A4; ++i) pin(ra[j][i]);
#pragma unroll
            for (int i = 0; i < B4; ++i) pin(rb[j][i]);
        }
        if (csm != 0.0f) {                                               // lane (l15, lq) holds rows m == lq (mod 4) of column l15
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) {
                float v = cs[jj];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                const int n = n0 + (wn * TN + jj) * 16 + l15;
                if (lq == 0 && n < g.N) atomicAdd(g.colsum + n, v);
            }
        }
      }
    } else
    if (nchunks > 0) {
        // prologue: chunk 0 -> LDS[0]; chunk 1 in flight in slot 1.  (Chunks beyond the split load zeros.)
        issue(0, ra[0], rb[0]);
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
                    land((c + 1) & 1, ra[(j + 1) % PF], rb[(j + 1) % PF]);
                    compute(c & 1, BR / 8, BR / 4);
                    __syncthreads();
                    ++c;
                    more = c < nchunks;
                }
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

// ---- the 128x128 tile with LDS-DMA staging -----------------------------------------------------------------------------
// Same contraction, same 32-row chunks, but the chunk goes global -> LDS by `buffer_load_dwordx4 ... lds` (no staging
// registers, no ds_write pass, no address VALU in the loop: the register-staged kernel above parks its 64-register ring in
// AGPRs and pays ~110 v_accvgpr moves per chunk for it).  An LDS-DMA wave-instruction writes 1 KiB lane-linearly, i.e. two
// unpadded 128-float rows; with unpadded rows the b32 fragment reads would conflict, so the fragments are read as ONE
// ds_read_b128 per operand per k-step instead: lane (l15, lq) takes floats [4*l15, 4*l15+4) of row m = 4*ks + lq, and
// accumulator (i, j) of the wave therefore holds output rows 4*rho + i (rho = the MFMA tile row) and columns 4*l15 + j --
// a permutation of which MFMA computes which output, free because the outputs are independent; the reduction order per
// output is what it was.  Two LDS buffers, one barrier per chunk: chunk c+1 is in flight while chunk c is multiplied.
template <bool GATHER, int BR, int NB>
__global__ __launch_bounds__(256) void gemm_tn_dma_kernel(const TnKArgs g)
{
    constexpr int BMo = 128, BNo = 128, TM = 4, TN = 4;
    constexpr int KS = BR / 4, NM = KS * 16;
    constexpr int PCS = BR / 8;                        // DMA pieces per operand per wave: 4 waves x 2 rows each
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                    // [NB][BR][BMo]
    float* Bs = smem + NB * BR * BMo;    // [NB][BR][BNo]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, lq = lane >> 4;
    const int ntn = (g.N + BNo - 1) / BNo;
    int k0, n0, mbeg;
    // Three workgroups share a CU and the hardware issues the OLDEST wave first: the oldest workgroup of a CU runs ahead, the
    // three "age ranks" of an XCD drift apart by more than its L2 holds, and the A panels all of them need are fetched once per
    // rank (dWout: 946 MB of fetches against 522 for the tile order's ideal).  prio_rot: every wave changes its issue priority
    // each chunk, cycling 0, 1, 2 with a phase taken from the XCD-local workgroup index / 32 (co-resident workgroups differ in
    // it under round-robin placement) -- no rank is favoured for long.
    int prio_ph = (int)((blockIdx.x >> 3) >> 5) % 3, prio_cnt = 0;     // prio_rot = chunks per priority phase
    if (g.xmap) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int nrow = (g.Kout + BMo - 1) / BMo;
        const int npr = ntn * g.splits;
        const int pr = xcd * ((npr + 7) >> 3) + slot / nrow;          // a contiguous, slab-major range of pairs per XCD (see gemm_tn_kernel)
        if (slot / nrow >= ((npr + 7) >> 3) || pr >= npr) return;
        k0 = (slot % nrow) * BMo; n0 = (pr % ntn) * BNo; mbeg = (pr / ntn) * g.mper;
    } else {
        k0 = (blockIdx.x / ntn) * BMo; n0 = (blockIdx.x % ntn) * BNo; mbeg = blockIdx.y * g.mper;
    }
    const int mend = (mbeg + g.mper < g.Mred) ? mbeg + g.mper : g.Mred;
    const int nchunks = mend > mbeg ? (mend - mbeg + BR - 1) / BR : 0;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // piece p of this wave: rows 8p + 2*wave + (lane >> 5) of the chunk, floats 4*(lane & 31) .. +3 of the tile row
    const int prow = 2 * wave + (lane >> 5), pcol = (lane & 31) * 4;
    const bool aok = k0 + pcol < g.Kout, bok = n0 + pcol < g.N;
    const uint32_t acol = (uint32_t)(k0 + pcol) * 4u, bcol = (uint32_t)(n0 + pcol) * 4u;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // gathered A: the row indices of a chunk are loaded one iteration before its pieces are issued (an index load waited for
    // inside the MFMA stream would drain the DMA queue with it: vmcnt retires in order)
    auto load_idx = [&](int c, int (&gi)[PCS]) __attribute__((always_inline)) {
#pragma unroll
        for (int P = 0; P < PCS; ++P) {
            const int m = mbeg + c * BR + P * 8 + prow;
            gi[P] = (GATHER && m < mend) ? g.rowidx[m] : 0;
        }
    };
    auto issue_piece = [&](int c, int buf, auto p_, const int (&gi)[PCS]) __attribute__((always_inline)) {
        constexpr int P = decltype(p_)::value;
        const int m0 = mbeg + c * BR;                 // a chunk past the slab: every lane out of range, zeros into the idle buffer
        if constexpr (P < PCS) {
            const int row = P * 8 + prow;
            const bool in = aok && m0 + row < mend;
            uint32_t vo;
            __amdgpu_buffer_rsrc_t rs;
            if constexpr (GATHER) {
                rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, (int)kOob, 0x00020000);
                vo = (uint32_t)gi[P] * (uint32_t)g.lda * 4u + acol;
            } else {
                rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A + (size_t)m0 * g.lda), 0, (int)kOob, 0x00020000);
                vo = (uint32_t)row * (uint32_t)g.lda * 4u + acol;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(As + (buf * BR + P * 8 + 2 * wave) * BMo), 16, in ? vo : kOob, 0, 0, 0);
        } else {
            constexpr int Q = P - PCS;
            const int row = Q * 8 + prow;
            const bool in = bok && m0 + row < mend;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B + (size_t)m0 * g.ldb), 0, (int)kOob, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(Bs + (buf * BR + Q * 8 + 2 * wave) * BNo), 16,
                                                     in ? (uint32_t)row * (uint32_t)g.ldb * 4u + bcol : kOob, 0, 0, 0);
        }
    };
    auto splice = [](int p) constexpr { return ((2 * p + 1) * (NM / 2)) / (2 * 2 * PCS); };   // pieces spread over the first half
    const float csm = (g.colsum && k0 == 0 && wm == 0) ? 1.0f : 0.0f;
    float cs[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) cs[j] = 0.f;

    if (nchunks > 0) {
        int gcur[PCS], gnxt[PCS];
        static_for<0, NB - 1>([&](auto c_) {
            load_idx(decltype(c_)::value, gcur);
            static_for<0, 2 * PCS>([&](auto p_) { issue_piece(decltype(c_)::value, decltype(c_)::value, p_, gcur); });
        });
        load_idx(NB - 1, gcur);
        wait_vmcnt<(NB - 2) * 2 * PCS>();
        __builtin_amdgcn_s_barrier();
        auto main_loop = [&](auto cs_) __attribute__((always_inline)) {
            constexpr bool CS = decltype(cs_)::value;
            int buf = 0, nbuf = NB - 1;                                  // the buffer multiplied / the one chunk c+NB-1 goes to
            for (int c = 0; c < nchunks; ++c) {
                if (g.prio_rot && --prio_cnt <= 0) {                      // (uniform; s_setprio takes an immediate)
                    prio_cnt = g.prio_rot;
                    if (prio_ph == 0) __builtin_amdgcn_s_setprio(0);
                    else if (prio_ph == 1) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(2);
                    prio_ph = prio_ph == 2 ? 0 : prio_ph + 1;
                }
                const f32x4* a = reinterpret_cast<const f32x4*>(As + (buf * BR + lq) * BMo + wm * 64 + l15 * 4);
                const f32x4* b = reinterpret_cast<const f32x4*>(Bs + (buf * BR + lq) * BNo + wn * 64 + l15 * 4);
                f32x4 av[2], bv[2];
                load_idx(c + NB, gnxt);
                av[0] = a[0];
                bv[0] = b[0];
                static_for<0, NM>([&](auto n_) {
                    constexpr int n = decltype(n_)::value, ks = n / 16, r = n % 16, i = r / TN, j = r % TN;
                    if constexpr (r == 0 && ks + 1 < KS) {
                        av[(ks + 1) & 1] = a[(ks + 1) * BMo];            // (ks+1)*4 rows further: 4*BMo floats = BMo f32x4
                        bv[(ks + 1) & 1] = b[(ks + 1) * BNo];
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks & 1][i], bv[ks & 1][j], acc[i][j], 0, 0, 0);
                    if constexpr (CS && i == 0) cs[j] += bv[ks & 1][j];
                    static_for<0, 2 * PCS>([&](auto p_) {
                        constexpr int p = decltype(p_)::value;
                        if constexpr (splice(p) == n) issue_piece(c + NB - 1, nbuf, p_, gcur);   // no branch in the MFMA stream
                    });
                });
                wait_vmcnt<(NB - 2) * 2 * PCS>();                         // chunk c+1 has landed; the later ones stay in flight
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                nbuf = buf;
                buf = buf + 1 == NB ? 0 : buf + 1;
#pragma unroll
                for (int P = 0; P < PCS; ++P) gcur[P] = gnxt[P];
            }
        };
        if (csm != 0.0f) main_loop(std::true_type{});
        else main_loop(std::false_type{});
        wait_vmcnt<0>();                                                 // the zero chunks issued past the slab, too
        __builtin_amdgcn_s_barrier();
        if (csm != 0.0f) {                                               // lane (l15, lq) holds rows m == lq (mod 4) of columns 4*l15 + j
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float v = cs[j];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                const int n = n0 + wn * 64 + l15 * 4 + j;
                if (lq == 0 && n < g.N) atomicAdd(g.colsum + n, v);
            }
        }
    }

    // epilogue through LDS (the staging buffers, free once every wave's last DMA has landed): the accumulators hold 4
    // consecutive columns per lane, the atomics / stores want 64 consecutive columns per wave-instruction
    constexpr int RGN = NB * BR * 256 / 4 < 4096 ? NB * BR * 256 / 4 : 4096;      // floats per wave
    constexpr int IPP = RGN >= 4096 ? 4 : RGN >= 2048 ? 2 : 1;                    // accumulator rows i per pass (16 tile rows x 64 columns each)
    static_assert(IPP >= 1 && TM % IPP == 0, "epilogue passes");
    float* tr = smem + wave * RGN;
    const int n = n0 + wn * 64 + lane;
    const int kb = k0 + wm * 64;
#pragma unroll
    for (int h = 0; h < TM / IPP; ++h) {
        if (h > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // wave-private region: program order is enough
#pragma unroll
        for (int ii = 0; ii < IPP; ++ii)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = h * IPP + ii;
                *reinterpret_cast<f32x4*>(tr + ((lq * 4 + r) * IPP + ii) * 64 + l15 * 4) = f32x4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
            }
        if (n < g.N) {
#pragma unroll 8
            for (int lr = 0; lr < 16 * IPP; ++lr) {
                const int k = kb + (lr / IPP) * 4 + h * IPP + lr % IPP;
                if (k >= g.Kout) continue;
                const float v = tr[lr * 64 + lane];
                float* p = g.C + (size_t)k * g.ldc + n;
                if (g.atomic) atomicAdd(p, v);
                else if (g.accumulate) *p = *p + v;
                else *p = v;
            }
        }
    }
}

}  // namespace s2vt
