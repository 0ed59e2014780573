// gemm_mfma.h -- the one contraction kernel of the forward path, hand-written for gfx950.
//
//   C[m, n] = chain_{k ascending} A[row(m), k] * W[k, n]      (+ fused epilogue)
//
// * v_mfma_f32_16x16x4_f32: exact fp32, and bit-for-bit an ascending-k fmaf chain, so with no
//   split-K the result equals the CPU oracle's chain bit-for-bit.
// * A is a concatenation of up to three K-segments (the reference's tf.concat([x, h]) operand,
//   tf_s2vt.py:119-143), each optionally a row gather (tf.nn.embedding_lookup, :128-134) or a
//   row broadcast (row % rowmod: K samples of one video share an operand).  A zero segment
//   (the `padding` input) is simply absent: zero products leave an fmaf chain unchanged.
// * Epilogues: STORE (+bias, +tanh), LSTM (BasicLSTMCell pointwise + DropoutWrapper, i/j/f/o of a
//   unit live in the same lane because the tile takes the same 16 units from all 4 gate column
//   groups), PICK (vocab logits + Gumbel-max / argmax -> packed 64-bit atomicMax; the logits never
//   go to HBM).
// * Tiling: 64-lane waves, WM x WN waves per workgroup, TM x TN 16x16 accumulators per wave,
//   BK = 32 K-chunk (one 128-B line per A row), global -> registers -> LDS double buffer, one
//   barrier per chunk.  LDS images are bank-conflict-free for the ds_read_b32 fragment reads:
//   A rows are 34 floats apart (bank = 2*row + k), B rows are == 16 (mod 32) floats apart.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "detmath.h"

namespace s2vt {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int SA = BK + 2;

enum { EPI_STORE = 0, EPI_LSTM = 1, EPI_PICK = 2 };

struct ASeg {
    const float* ptr;    // [rows, ld] row-major; nullptr = segment absent
    const int* rowidx;   // optional gather: row(m) = rowidx[m]
    const unsigned long long* rowkey;  // optional gather through packed PICK results: row(m) = ~low32(rowkey[m])
    int ld;
    int k;               // segment length along K
    int kw;              // first row of W this segment multiplies
    int rowmod;          // >0: row(m) = m % rowmod (applied before rowidx)
};

struct GemmArgs {
    ASeg seg[3];
    int nseg;
    const float* W;      // [K_total, ldw]
    int ldw;
    int M;
    int N;               // columns per group (LSTM: H units per gate; otherwise the full width)
    int gstride;         // column distance between groups in W / bias (LSTM: H)
    const float* bias;   // [NG * gstride] or nullptr
    const float* cinit;  // optional initial accumulator [*, ldcinit] (a carried partial chain)
    int ldcinit;
    int cinit_rowmod;
    // EPI_STORE
    float* C;
    int ldc;
    int act;             // 0 none, 1 tanh
    int xcd_map;         // 1: XCD-aware tile order (set by the launcher for skinny-M shapes)
    int splits;          // >1: order-free split-K over blockIdx.y (backward data path only, nseg == 1)
    int kper;            //     K range per split (multiple of BK)
    size_t slab_stride;  //     floats between the partial-sum slabs of consecutive splits
    // EPI_LSTM
    const float* c_prev;
    int cprev_rowmod;    // >0: c_prev row = m % cprev_rowmod (K samples start from one encoder state)
    float* c_new;
    float* h_new;
    float* out;          // dropped output (== h_new values when keep >= 1)
    float* gates;        // optional [M, 4H] activated gates (si | tj | sf | so), saved for backward
    float keep;          // >= 1: no dropout
    uint32_t drop_code;
    // EPI_PICK (and dropout) noise stream
    const int* video_id;
    const int* sample_id;
    uint32_t seed_lo, seed_hi;
    int step;
    unsigned long long* pick;   // [M] packed (orderable(key) << 32) | ~index, zeroed before the launch
    float* logits_out;          // optional [M, ldc]
};

// ---- loads the compiler does not schedule (cdna_hip_programming.md §5.7): hipcc sinks ordinary prefetch
// loads next to their first use and drains them with vmcnt(0); issued as asm they stay where they are
// written, and the ring is drained with a hand-counted s_waitcnt below.
__device__ __forceinline__ void gload16(f32x4& d, const float* p)
{
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void pin(f32x4& v) { asm volatile("" : "+v"(v)); }

template <int WM, int WN, int TM, int TN, int NG, int EPI, bool VEC>
struct GemmCfg {
    static constexpr int NT = 64 * WM * WN;
    static constexpr int BM = WM * TM * 16;
    static constexpr int BN = WN * TN * 16;
    static constexpr int TNG = TN / NG;            // subtiles per group per wave
    static constexpr int CG = WN * TNG * 16;       // tile columns per group
    static constexpr int SB = (BN % 32 == 16) ? BN : BN + 16;
    static constexpr int A4 = (BM * (BK / 4) + NT - 1) / NT;   // float4 per thread per chunk
    static constexpr int B4 = (BK * (BN / 4) + NT - 1) / NT;
    static constexpr int LDS_FLOATS = 2 * (BM * SA + BK * SB);
    // prefetch ring depth (chunks in flight per thread): the skinny-M kernels are bound by operand bytes
    // in flight per CU (measured: ~3500-cycle loaded latency), so the ring is as deep as ~96 staging
    // VGPRs allow, between 2 and 6 slots.
#ifndef S2VT_LAND_AT
#define S2VT_LAND_AT 8   /* k-steps of a chunk computed before the next chunk is landed in LDS (8 = after all) */
#endif
#ifndef S2VT_PF_BUDGET
#define S2VT_PF_BUDGET 48
#endif
    static constexpr int PF_RAW = S2VT_PF_BUDGET / (4 * (A4 + B4));
    static constexpr int PF = PF_RAW < 2 ? 2 : (PF_RAW > 6 ? 6 : PF_RAW);
    static_assert(TN % NG == 0, "TN must split evenly over the column groups");
};

template <int WM, int WN, int TM, int TN, int NG, int EPI, bool VEC>
__global__ __launch_bounds__(64 * WM * WN) void gemm_kernel(const GemmArgs g)
{
    using Cfg = GemmCfg<WM, WN, TM, TN, NG, EPI, VEC>;
    constexpr int NT = Cfg::NT, BM = Cfg::BM, BN = Cfg::BN, TNG = Cfg::TNG, CG = Cfg::CG, SB = Cfg::SB;
    constexpr int A4 = Cfg::A4, B4 = Cfg::B4;
    constexpr int PF = Cfg::PF;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][BM][SA]
    float* Bs = smem + 2 * BM * SA;         // [2][BK][SB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, lq = lane >> 4;

    // Tile order.  Workgroups are dealt round-robin over the 8 XCDs (b % 8 labels the blocks that share
    // an L2).  For skinny-M shapes (few row tiles) every row tile of one COLUMN tile is placed on the
    // same XCD, back to back: the weight panel W[:, tile] -- the big operand, read once per row tile --
    // is then fetched from HBM / Infinity Cache into ONE L2 instead of into up to 8 (measured: L2 miss
    // traffic 400 MB -> 70 MB per LSTM2 launch).  Placement is a speed choice only.
    const int ntile_n = (g.N + CG - 1) / CG;
    const int ntile_m = (g.M + BM - 1) / BM;
    int tile_m, tile_n;
    if (g.xcd_map) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        tile_m = slot % ntile_m;
        tile_n = (slot / ntile_m) * 8 + xcd;
        if (tile_n >= ntile_n) return;
    } else {
        tile_n = blockIdx.x % ntile_n;
        tile_m = blockIdx.x / ntile_n;
    }
    const int m0 = tile_m * BM;
    const int n0 = tile_n * CG;             // within-group column offset

    f32x4 acc[TM][TN];
    // initial accumulator: +0 or a carried partial chain
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (g.cinit) {
                const int col = (j / TNG) * g.gstride + n0 + (wn * TNG + j % TNG) * 16 + l15;
                const bool cok = n0 + (wn * TNG + j % TNG) * 16 + l15 < g.N;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int m = m0 + (wm * TM + i) * 16 + lq * 4 + r;
                    if (m < g.M && cok) {
                        if (g.cinit_rowmod > 0) m %= g.cinit_rowmod;
                        v[r] = g.cinit[(size_t)m * g.ldcinit + col];
                    }
                }
            }
            acc[i][j] = v;
        }

    // per-thread staging ring: PF chunks in flight between HBM/L2 and the LDS double buffer
    f32x4 ra[PF][A4], rb[PF][B4];
    unsigned pa[PF], pb[PF];       // validity bits of the ring slots (zero-fill happens when a chunk lands)
    constexpr int LPC = A4 + B4;                                   // asm-issued loads per chunk per thread
    constexpr int WAITN = ((PF - 1) * LPC > 63) ? 63 : (PF - 1) * LPC;

    // ---- wave-uniform description of the K walk: chunk c -> (segment, offset) by scalar arithmetic
    const int kbeg = g.splits > 1 ? (int)blockIdx.y * g.kper : 0;
    int slen[3], nch[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        int k = 0;
        if (i < g.nseg && g.seg[i].ptr != nullptr) {
            k = g.seg[i].k - kbeg;
            if (g.splits > 1 && k > g.kper) k = g.kper;
            if (k < 0) k = 0;
        }
        slen[i] = k;
        nch[i] = (k + BK - 1) / BK;
    }
    const int cum0 = nch[0], cum1 = nch[0] + nch[1], nchunks = nch[0] + nch[1] + nch[2];

    // Row offsets of this thread's A slots for every segment, resolved ONCE (gather / broadcast index
    // loads happen here, never inside the pipelined loop).  -1 marks a row beyond M / an absent segment.
    int aoff0[A4], aoff1[A4], aoff2[A4];
    auto row_offsets = [&](int sidx, int (&ao)[A4]) {
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            const int idx = tid + i * NT;
            int m = m0 + idx / (BK / 4);
            int off = -1;
            if (slen[sidx] > 0 && idx < BM * (BK / 4) && m < g.M) {
                const ASeg& sg = g.seg[sidx];
                if (sg.rowmod > 0) m %= sg.rowmod;
                if (sg.rowidx) m = sg.rowidx[m];
                if (sg.rowkey) m = (int)(~(uint32_t)sg.rowkey[m]);
                off = m * sg.ld + kbeg;
            }
            ao[i] = off;
        }
    };
    row_offsets(0, aoff0);
    row_offsets(1, aoff1);
    row_offsets(2, aoff2);

    // Issue the global loads of chunk c into a ring slot.  UNCONDITIONAL and always safe: addresses are
    // clamped into the segment; the validity bits zero the out-of-range elements when the chunk lands.
    // A chunk index beyond the walk yields an all-zero chunk.
    auto issue = [&](int c, f32x4 (&qa)[A4], f32x4 (&qb)[B4], unsigned& ma, unsigned& mb) {
        const int cc_ = c < nchunks ? c : nchunks - 1;
        const int sidx = (cc_ >= cum0 ? 1 : 0) + (cc_ >= cum1 ? 1 : 0);
        const int cstart = sidx == 0 ? 0 : (sidx == 1 ? cum0 : cum1);
        const int koff = (c - cstart) * BK;
        const float* abase = sidx == 0 ? g.seg[0].ptr : (sidx == 1 ? g.seg[1].ptr : g.seg[2].ptr);
        const int sk = sidx == 0 ? slen[0] : (sidx == 1 ? slen[1] : slen[2]);
        const int kw = (sidx == 0 ? g.seg[0].kw : (sidx == 1 ? g.seg[1].kw : g.seg[2].kw)) + kbeg;
        unsigned va = 0, vb = 0;
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            const int idx = tid + i * NT;
            const int k = koff + (idx % (BK / 4)) * 4;
            const int ro = sidx == 0 ? aoff0[i] : (sidx == 1 ? aoff1[i] : aoff2[i]);
            if constexpr (VEC) {
                const bool ok = ro >= 0 && k < sk;
                gload16(qa[i], abase + (ok ? ro + k : 0));
                va |= (ok ? 1u : 0u) << i;
            } else {   // odd shapes (tests): ordinary compiler-scheduled loads, zero-filled right here
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool ok = ro >= 0 && k + e < sk;
                    const float x = abase[ok ? ro + k + e : 0];
                    v[e] = ok ? x : 0.f;
                }
                qa[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < B4; ++i) {
            const int idx = tid + i * NT;
            const int kr = idx / (BN / 4);
            const int col = (idx % (BN / 4)) * 4;
            const int grp = col / CG, cc = n0 + col % CG;
            const int k = koff + kr;
            const bool kok = (B4 * NT == BK * (BN / 4) || idx < BK * (BN / 4)) && k < sk;
            const float* wrow = g.W + (size_t)(kw + (kok ? k : 0)) * g.ldw + grp * g.gstride;
            if constexpr (VEC) {
                const bool ok = kok && cc < g.N;
                gload16(qb[i], wrow + (ok ? cc : 0));
                vb |= (ok ? 1u : 0u) << i;
            } else {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool ok = kok && cc + e < g.N;
                    const float x = wrow[ok ? cc + e : 0];
                    v[e] = ok ? x : 0.f;
                }
                qb[i] = v;
            }
        }
        ma = va;
        mb = vb;
    };

    // Land a ring slot in an LDS buffer.  The caller has already waited (hand-counted vmcnt) for this
    // slot's loads; pin() keeps every consumer below that wait.
    auto land = [&](int buf, f32x4 (&qa)[A4], f32x4 (&qb)[B4], unsigned ma, unsigned mb) {
        float* a = As + buf * BM * SA;
        float* b = Bs + buf * BK * SB;
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            if constexpr (VEC) pin(qa[i]);
            f32x4 v = qa[i];
            if constexpr (VEC)
                if (!((ma >> i) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
            const int idx = tid + i * NT;
            if (A4 * NT == BM * (BK / 4) || idx < BM * (BK / 4)) {
                float* d = a + (idx / (BK / 4)) * SA + (idx % (BK / 4)) * 4;
                *reinterpret_cast<float2*>(d) = make_float2(v[0], v[1]);
                *reinterpret_cast<float2*>(d + 2) = make_float2(v[2], v[3]);
            }
        }
#pragma unroll
        for (int i = 0; i < B4; ++i) {
            if constexpr (VEC) pin(qb[i]);
            f32x4 v = qb[i];
            if constexpr (VEC)
                if (!((mb >> i) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
            const int idx = tid + i * NT;
            if (B4 * NT == BK * (BN / 4) || idx < BK * (BN / 4)) {
                float* d = b + (idx / (BN / 4)) * SB + (idx % (BN / 4)) * 4;
                *reinterpret_cast<f32x4*>(d) = v;
            }
        }
    };

    // MFMAs of k-steps [KS0, KS1) of one chunk (hipcc interleaves the fragment reads with the MFMAs).
    auto compute = [&](int buf, auto ks0_, auto ks1_) {
        constexpr int KS0 = decltype(ks0_)::value, KS1 = decltype(ks1_)::value;
        const float* a = As + buf * BM * SA + ((wm * TM) * 16 + l15) * SA + lq;
        const float* b = Bs + buf * BK * SB + lq * SB + l15;
#pragma unroll
        for (int ks = KS0; ks < KS1; ++ks) {
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = a[i * 16 * SA + ks * 4];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = b[ks * 4 * SB + (j / TNG) * CG + (wn * TNG + j % TNG) * 16];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };
    using K0 = std::integral_constant<int, 0>;
    using KH = std::integral_constant<int, S2VT_LAND_AT>;
    using K8 = std::integral_constant<int, BK / 4>;

    if (nchunks > 0) {
        // prologue: chunk 0 -> LDS[0]; chunks 1 .. PF-1 in flight in ring slots 1 .. PF-1
        issue(0, ra[0], rb[0], pa[0], pb[0]);
        if constexpr (VEC) wait_vmcnt<0>();
        land(0, ra[0], rb[0], pa[0], pb[0]);
#pragma unroll
        for (int j = 1; j < PF; ++j) issue(j, ra[j], rb[j], pa[j], pb[j]);
        __syncthreads();

        // Steady state: iteration c has LDS[c&1] = chunk c and ring slot (c+i)%PF = chunk c+i in flight
        // (i = 1..PF-1).  It issues chunk c+PF into the slot chunk c came from, computes the first
        // S2VT_LAND_AT k-steps of chunk c, waits until all but the youngest (PF-1) chunks' loads have
        // returned (vmcnt is in-order), lands chunk c+1 in the OTHER LDS buffer (nobody reads it between
        // the previous barrier and the next one) so that the LDS stores retire under the remaining MFMAs,
        // computes the rest, one barrier.  Unrolled by PF so ring slots are compile-time constants; the
        // only branch is the wave-uniform loop exit.
        int c = 0;
        bool more = true;
        while (more) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                if (more) {
                    issue(c + PF, ra[j], rb[j], pa[j], pb[j]);
                    compute(c & 1, K0{}, KH{});
                    if constexpr (VEC) wait_vmcnt<WAITN>();
                    land((c + 1) & 1, ra[(j + 1) % PF], rb[(j + 1) % PF], pa[(j + 1) % PF], pb[(j + 1) % PF]);
                    compute(c & 1, KH{}, K8{});
                    __syncthreads();
                    ++c;
                    more = c < nchunks;
                }
            }
        }
        if constexpr (VEC) wait_vmcnt<0>();     // the ring's last PF-1 (all-zero, beyond-the-walk) chunks: drain before the epilogue
    }

    // ------------------------------------------------------------------ epilogues
    if constexpr (EPI == EPI_STORE) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cc = n0 + (wn * TNG + j % TNG) * 16 + l15;
            const int col = (j / TNG) * g.gstride + cc;
            if (cc >= g.N) continue;
            const float bj = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + (wm * TM + i) * 16 + lq * 4 + r;
                    if (m < g.M) {
                        float v = acc[i][j][r];
                        if (g.bias) v = v + bj;
                        if (g.act == 1) v = dm_tanhf(v);
                        g.C[(size_t)blockIdx.y * g.slab_stride + (size_t)m * g.ldc + col] = v;
                    }
                }
        }
    } else if constexpr (EPI == EPI_LSTM) {
        static_assert(EPI != EPI_LSTM || NG == 4, "LSTM epilogue needs the four gate groups");
        const int H = g.N;
#pragma unroll
        for (int jj = 0; jj < TNG; ++jj) {
            const int u = n0 + (wn * TNG + jj) * 16 + l15;
            if (u >= H) continue;
            const float bi = g.bias[u], bj = g.bias[H + u], bf = g.bias[2 * H + u], bo = g.bias[3 * H + u];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + (wm * TM + i) * 16 + lq * 4 + r;
                    if (m >= g.M) continue;
                    const float zi = acc[i][0 * TNG + jj][r] + bi;
                    const float zj = acc[i][1 * TNG + jj][r] + bj;
                    const float zf = acc[i][2 * TNG + jj][r] + bf;
                    const float zo = acc[i][3 * TNG + jj][r] + bo;
                    const float si = dm_sigmoidf(zi);
                    const float tj = dm_tanhf(zj);
                    const float sf = dm_sigmoidf(zf + 1.0f);
                    const float so = dm_sigmoidf(zo);
                    const size_t o = (size_t)m * H + u;
                    const size_t op = (size_t)(g.cprev_rowmod > 0 ? m % g.cprev_rowmod : m) * H + u;
                    const float t1 = g.c_prev[op] * sf;
                    const float t2 = si * tj;
                    const float c = t1 + t2;
                    const float h = dm_tanhf(c) * so;
                    g.c_new[o] = c;
                    g.h_new[o] = h;
                    if (g.out) {
                        float ov = h;
                        if (g.keep < 1.0f) {
                            const float k01 = dropout_keep01(g.seed_lo, g.seed_hi, (uint32_t)g.video_id[m],
                                                             (uint32_t)g.sample_id[m], g.drop_code, (uint32_t)u, g.keep);
                            ov = (h / g.keep) * k01;
                        }
                        g.out[o] = ov;
                    }
                    if (g.gates) {
                        float* gp = g.gates + (size_t)m * 4 * H + u;
                        gp[0] = si; gp[H] = tj; gp[2 * H] = sf; gp[3 * H] = so;
                    }
                }
        }
    } else {  // EPI_PICK
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + (wm * TM + i) * 16 + lq * 4 + r;
                const bool mok = m < g.M;
                int sid = -1, vid = 0;
                if (mok) { sid = g.sample_id[m]; vid = g.video_id[m]; }
                float best = 0.0f;
                uint32_t bidx = 0xFFFFFFFFu;
                bool have = false;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = n0 + (wn * TN + j) * 16 + l15;
                    if (mok && col < g.N) {
                        float v = acc[i][j][r] + g.bias[col];
                        if (g.logits_out) g.logits_out[(size_t)m * g.ldc + col] = v;
                        if (sid >= 0)
                            v = v + gumbel_at(g.seed_lo, g.seed_hi, (uint32_t)vid, (uint32_t)sid, (uint32_t)g.step,
                                              (uint32_t)col);
                        v = v + 0.0f;  // -0 -> +0 so that the integer order equals the float order
                        if (!have || v > best) { best = v; bidx = (uint32_t)col; have = true; }
                    }
                }
                // reduce over the 16 lanes that hold this row's columns (ties -> lowest index)
                unsigned long long key = have ? (((unsigned long long)orderable(best) << 32) | (uint32_t)(~bidx)) : 0ull;
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    const unsigned long long o = __shfl_xor(key, off, 64);
                    key = o > key ? o : key;
                }
                if (l15 == 0 && mok && key != 0ull) atomicMax(&g.pick[m], key);
            }
    }
}

}  // namespace s2vt
