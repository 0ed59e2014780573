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

enum { EPI_STORE = 0, EPI_LSTM = 1, EPI_PICK = 2, EPI_LSTM_GW = 3 };

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
    int dbg;             // S2VT_ABLATE builds only: timing ablation bits (1 no MFMA, 2 no LDS stores, 4 no global loads, 8 no barrier, 16 no fragment reads)
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
// Raw-buffer form of the ring load: address = descriptor base + per-lane byte offset + wave-uniform byte offset, and
// a lane whose offset is >= the descriptor's num_records gets ZEROS back without touching memory -- so a chunk
// element out of range (row >= M, k beyond the segment, column >= N) costs a compare + select of the OFFSET (or
// nothing at all, when the condition is loop-invariant) instead of a 64-bit address computation and select.
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kOob = 0x80000000u;                       // = num_records of every descriptor built here
__device__ __forceinline__ i32x4 make_rsrc(const float* base)
{
    const uint64_t u = reinterpret_cast<uint64_t>(base);
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)u);
    r[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(u >> 32) & 0xffffu));   // stride 0, no swizzle
    r[2] = (int)kOob;                                                             // num_records (bytes)
    r[3] = 0x00020000;                                                            // gfx9 raw buffer, 32-bit elements
    return r;
}
__device__ __forceinline__ void bload16(f32x4& d, uint32_t voff, i32x4 rsrc, uint32_t soff)
{
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(d) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void pin(f32x4& v) { asm volatile("" : "+v"(v)); }
// 16 zero bytes in device memory: an out-of-range lane of a ring load reads THESE instead of a clamped
// in-range address, so its chunk element arrives as zeros and needs no select when it lands in LDS
static __device__ const float s2vt_zero16[4] __attribute__((aligned(16), used)) = {0.f, 0.f, 0.f, 0.f};
// a wave-uniform value as an opaque SGPR value (v_readfirstlane): the optimizer cannot look through it
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const float* uniform(const float* p)
{
    const uint64_t u = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
    return reinterpret_cast<const float*>(((uint64_t)hi << 32) | lo);
}

#ifdef S2VT_STAMP
// Dev build only (tools/stamp_loop.py): per-segment shader-clock sums of the main loop, one s_memtime per boundary.
static __device__ unsigned long long s2vt_stamp_acc[16];
__device__ __forceinline__ unsigned long long stamp_now()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define S2VT_STAMP_AT(i) do { const unsigned long long t_ = stamp_now(); st_acc[i] += t_ - st_last; st_last = t_; } while (0)
#else
#define S2VT_STAMP_AT(i)
#endif

template <int WM, int WN, int TM, int TN, int NG, int EPI, bool VEC, int NBUF = 2, int BKT = 32>
struct GemmCfg {
    // K-chunk depth of this configuration (k per barrier): 32 by default; 64 / 128 for the tiles whose chunk holds few
    // MFMAs per wave (M = 64 step kernels: 8 per chunk at 32), where the per-chunk barrier and waits dominate
    static constexpr int BK = BKT;
    static constexpr int SA = BKT + 2;                        // A rows BK + 2 floats apart: bank = 2*row + k
    static constexpr int NT = 64 * WM * WN;
    static constexpr int BM = WM * TM * 16;
    static constexpr int BN = WN * TN * 16;
    // EPI_LSTM_GW ("gate per wave"): the four waves along N each take ONE gate column group of the same
    // TN*16 hidden units (WN == 4 == NG), so a workgroup is BM rows x TN*16 units and the grid can be cut
    // to ~one workgroup per CU for any M; the gates of a unit meet through LDS in the epilogue.
    static constexpr bool GW = (EPI == EPI_LSTM_GW);
    static constexpr int TNG = GW ? TN : TN / NG;            // subtiles per group per wave
    static constexpr int CG = GW ? TN * 16 : WN * TNG * 16;  // tile columns per group
    static constexpr int ZS = 4 * CG + 4;                    // GW gate-exchange image: floats per row
    static constexpr int SB = (BN % 32 == 16) ? BN : BN + 16;
    static constexpr int A4 = (BM * (BK / 4) + NT - 1) / NT;   // float4 per thread per chunk
    static constexpr int B4 = (BK * (BN / 4) + NT - 1) / NT;
    static constexpr int LOOP_FLOATS = NBUF * (BM * SA + BK * SB);
    static constexpr int LDS_FLOATS = (GW && BM * ZS > LOOP_FLOATS) ? BM * ZS : LOOP_FLOATS;
    // Prefetch ring depth (chunks in flight per thread).  The skinny-M kernels are bound by operand bytes in
    // flight per CU: a chunk of a small tile is computed in ~0.2 us while a load takes ~2 us under load, so the
    // ring must hold ~10 chunks to cover it (measured: 16x16u tile, 4 chunks in flight: the loads cost 30 of
    // 71 us).  Small tiles have the registers for that (few accumulators); big tiles run two workgroups per CU
    // and keep the shallow ring.
#ifndef S2VT_LAND_AT
#define S2VT_LAND_AT 4   /* k-steps of a chunk computed before the next chunk is landed in LDS (8 = after all) */
#endif
#ifndef S2VT_PF_BUDGET
#define S2VT_PF_BUDGET 48
#endif
#ifndef S2VT_PF_BUDGET_SKINNY
#define S2VT_PF_BUDGET_SKINNY 48
#endif
#ifndef S2VT_PF_MAX_SKINNY
#define S2VT_PF_MAX_SKINNY 6
#endif
    static constexpr bool SKINNY = TM * TN <= 6;
    static constexpr int PF_RAW = (SKINNY ? S2VT_PF_BUDGET_SKINNY : S2VT_PF_BUDGET) / (4 * (A4 + B4));
    static constexpr int PF_CAP = SKINNY ? S2VT_PF_MAX_SKINNY : 6;
    static constexpr int PF2 = PF_RAW < 2 ? 2 : (PF_RAW > PF_CAP ? PF_CAP : PF_RAW);
    // NBUF == 3 (fragment-pipelined loop): the ring depth is even so that the fragment register set of a
    // chunk (chunk & 1) is a compile-time index inside the loop unrolled by PF
    static constexpr int PF = NBUF == 3 ? (PF2 & ~1) : PF2;
    static_assert(NBUF == 2 || NBUF == 3, "two or three LDS stages");
    static_assert(GW || TN % NG == 0, "TN must split evenly over the column groups");
    static_assert(!GW || (WN == 4 && NG == 4), "gate-per-wave needs four waves along N");
};

template <int WM, int WN, int TM, int TN, int NG, int EPI, bool VEC, int NBUF = 2, int BKT = 32>
__global__ __launch_bounds__(64 * WM * WN) void gemm_kernel(const GemmArgs g)
{
    using Cfg = GemmCfg<WM, WN, TM, TN, NG, EPI, VEC, NBUF, BKT>;
    constexpr int BK = Cfg::BK, SA = Cfg::SA;                  // (shadow the namespace-scope defaults)
    constexpr int NT = Cfg::NT, BM = Cfg::BM, BN = Cfg::BN, TNG = Cfg::TNG, CG = Cfg::CG, SB = Cfg::SB;
    constexpr int A4 = Cfg::A4, B4 = Cfg::B4;
    constexpr int PF = Cfg::PF;

#ifdef S2VT_STAMP
    unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = stamp_now();
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [NBUF][BM][SA]
    float* Bs = smem + NBUF * BM * SA;      // [NBUF][BK][SB]

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
        // Many-tile shapes: XCD b % 8 takes a contiguous band of the tile sequence (column tiles fastest), so the
        // workgroups that share an A row block run on ONE XCD and it is fetched into one L2 instead of eight
        // (bijective for any tile count).
        // (only for wide outputs, >= 16 column tiles: with few column tiles the plain order already gives every
        // XCD its own one or two weight panels, and banding would re-stream the whole weight matrix per row block
        // -- measured 1.0 GB -> 1.9 GB of fabric reads on the 8000x4000x1500 data-gradient product)
        int t = blockIdx.x;
        if (ntile_n >= 16) {
            const int nwg = ntile_m * ntile_n, q = nwg >> 3, r = nwg & 7;
            const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
            t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
        }
        tile_n = t % ntile_n;
        tile_m = t / ntile_n;
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
                const int cu = Cfg::GW ? n0 + j * 16 + l15 : n0 + (wn * TNG + j % TNG) * 16 + l15;   // column within its group
                const int col = (Cfg::GW ? wn : j / TNG) * g.gstride + cu;
                const bool cok = cu < g.N;
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
    // The three segment descriptors as named scalars: a run-time index into the by-value argument struct
    // makes hipcc copy the struct to scratch, and scratch loads share vmcnt with the ring (measured: every
    // chunk then drains the whole prefetch ring, 2.2 us per chunk).
    auto seg_len = [&](const float* ptr, int sk, int i) __attribute__((always_inline)) {
        int k = 0;
        if (i < g.nseg && ptr != nullptr) {
            k = sk - kbeg;
            if (g.splits > 1 && k > g.kper) k = g.kper;
            if (k < 0) k = 0;
        }
        return k;
    };
    // (uniform() makes each field an opaque SGPR value: left alone, hipcc turns "sidx == 0 ? seg[0].f : ..."
    // into a load through a selected ADDRESS, which forces the by-value argument struct into scratch)
    const float* const sp0 = uniform(g.seg[0].ptr);
    const float* const sp1 = uniform(g.seg[1].ptr);
    const float* const sp2 = uniform(g.seg[2].ptr);
    const int skw0 = uniform(g.seg[0].kw), skw1 = uniform(g.seg[1].kw), skw2 = uniform(g.seg[2].kw);
    const int slen0 = uniform(seg_len(sp0, g.seg[0].k, 0)), slen1 = uniform(seg_len(sp1, g.seg[1].k, 1)),
              slen2 = uniform(seg_len(sp2, g.seg[2].k, 2));
    const int nch0 = (slen0 + BK - 1) / BK, nch1 = (slen1 + BK - 1) / BK, nch2 = (slen2 + BK - 1) / BK;
    const int cum0 = nch0, cum1 = nch0 + nch1, nchunks = nch0 + nch1 + nch2;

    // Row offsets of this thread's A slots for every segment, resolved ONCE (gather / broadcast index
    // loads happen here, never inside the pipelined loop).  -1 marks a row beyond M / an absent segment.
    int aoff0[A4], aoff1[A4], aoff2[A4];
    auto row_offsets = [&](int sl, int rowmod, const int* rowidx, const unsigned long long* rowkey, int ld, int (&ao)[A4]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            const int idx = tid + i * NT;
            int m = m0 + idx / (BK / 4);
            int off = -1;
            if (sl > 0 && idx < BM * (BK / 4) && m < g.M) {
                if (rowmod > 0) m %= rowmod;
                if (rowidx) m = rowidx[m];
                if (rowkey) m = (int)(~(uint32_t)rowkey[m]);
                off = m * ld + kbeg;
            }
            ao[i] = off;
        }
    };
    row_offsets(slen0, g.seg[0].rowmod, g.seg[0].rowidx, g.seg[0].rowkey, g.seg[0].ld, aoff0);
    row_offsets(slen1, g.seg[1].rowmod, g.seg[1].rowidx, g.seg[1].rowkey, g.seg[1].ld, aoff1);
    row_offsets(slen2, g.seg[2].rowmod, g.seg[2].rowidx, g.seg[2].rowkey, g.seg[2].ld, aoff2);

    // Issue the global loads of chunk c into a ring slot.  UNCONDITIONAL and always safe: an out-of-range
    // element (row >= M, k beyond the segment, column >= N, chunk beyond the walk) is loaded from
    // s2vt_zero16 instead, so it lands as zeros.
    // (The chunk -> segment selection is done by the S2VT_ISSUE macro at KERNEL scope, on plain local values:
    // inside a lambda the same "sidx == 0 ? a : b" picks between by-reference captures, which hipcc folds
    // into a load through a run-time offset into the closure object -- the closure, every local it points
    // to and the argument struct then live in scratch, and scratch loads share vmcnt with the ring.)
    auto issue_at = [&](int koff, const float* abase, int sk, int kw, const int (&aro)[A4], f32x4 (&qa)[A4], f32x4 (&qb)[B4],
                        unsigned& ma, unsigned& mb) __attribute__((always_inline)) {
#ifdef S2VT_ABLATE
        if (g.dbg & 4) { ma = 0; mb = 0; return; }
#endif
        unsigned va = 0, vb = 0;
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            const int idx = tid + i * NT;
            const int k = koff + (idx % (BK / 4)) * 4;
            const int ro = aro[i];
            if constexpr (VEC) {
                const bool ok = ro >= 0 && k < sk;
                gload16(qa[i], ok ? abase + (ro + k) : s2vt_zero16);
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
                gload16(qb[i], ok ? wrow + cc : s2vt_zero16);
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
#define S2VT_ISSUE(CHUNK, SLOT)                                                                          \
    do {                                                                                                 \
        const int c_ = (CHUNK);                                                                          \
        const int cc_ = c_ < nchunks ? c_ : nchunks - 1;                                                 \
        const int sidx_ = (cc_ >= cum0 ? 1 : 0) + (cc_ >= cum1 ? 1 : 0);                                 \
        const int cstart_ = sidx_ == 0 ? 0 : (sidx_ == 1 ? cum0 : cum1);                                 \
        const float* abase_ = sidx_ == 0 ? sp0 : (sidx_ == 1 ? sp1 : sp2);                               \
        const int sk_ = sidx_ == 0 ? slen0 : (sidx_ == 1 ? slen1 : slen2);                               \
        const int kw_ = (sidx_ == 0 ? skw0 : (sidx_ == 1 ? skw1 : skw2)) + kbeg;                         \
        int aro_[A4];                                                                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < A4; ++i_)                                                \
            aro_[i_] = sidx_ == 0 ? aoff0[i_] : (sidx_ == 1 ? aoff1[i_] : aoff2[i_]);                    \
        issue_at((c_ - cstart_) * BK, abase_, sk_, kw_, aro_, ra[SLOT], rb[SLOT], pa[SLOT], pb[SLOT]);   \
    } while (0)

    // Land a ring slot in an LDS buffer.  The caller has already waited (hand-counted vmcnt) for this
    // slot's loads; pin() keeps every consumer below that wait.
    auto land = [&](int buf, f32x4 (&qa)[A4], f32x4 (&qb)[B4], unsigned ma, unsigned mb) __attribute__((always_inline)) {
#ifdef S2VT_ABLATE
        if (g.dbg & 2) return;
#endif
        float* a = As + buf * BM * SA;
        float* b = Bs + buf * BK * SB;
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            if constexpr (VEC) pin(qa[i]);
            f32x4 v = qa[i];
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
            const int idx = tid + i * NT;
            if (B4 * NT == BK * (BN / 4) || idx < BK * (BN / 4)) {
                float* d = b + (idx / (BN / 4)) * SB + (idx % (BN / 4)) * 4;
                *reinterpret_cast<f32x4*>(d) = v;
            }
        }
    };

    // The ring's last PF-1 chunks (beyond the walk, never landed) are still in flight when the loop exits.
    // Wait for them AND keep every ring register formally alive until after that wait: an asm load's
    // destination is "written" at the asm statement as far as hipcc knows, so a slot that is never read again
    // is free at once, and hipcc hands it to epilogue values (pure register code may be scheduled above an
    // asm volatile) that the late-arriving load data then overwrites -- seen as wrong dwords / wild addresses
    // whenever the operands were cold in cache.
    auto drain_ring = [&]() __attribute__((always_inline)) {
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
    };

    // MFMAs of k-steps [KS0, KS1) of one chunk (hipcc interleaves the fragment reads with the MFMAs).
    auto compute = [&](int buf, auto ks0_, auto ks1_) __attribute__((always_inline)) {
        constexpr int KS0 = decltype(ks0_)::value, KS1 = decltype(ks1_)::value;
#ifdef S2VT_ABLATE
        if (g.dbg & 16) {
#pragma unroll
            for (int ks = KS0; ks < KS1; ++ks)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, 1.0f, acc[i][j], 0, 0, 0);
            return;
        }
        if (g.dbg & 1) {
            const float* a_ = As + buf * BM * SA + ((wm * TM) * 16 + l15) * SA + lq;
            const float* b_ = Bs + buf * BK * SB + lq * SB + l15;
            float sacc = 0.f;
#pragma unroll
            for (int ks = KS0; ks < KS1; ++ks) {
#pragma unroll
                for (int i = 0; i < TM; ++i) sacc += a_[i * 16 * SA + ks * 4];
#pragma unroll
                for (int j = 0; j < TN; ++j) sacc += b_[ks * 4 * SB + j * 16];
            }
            acc[0][0][0] += sacc;
            return;
        }
#endif
        const float* a = As + buf * BM * SA + ((wm * TM) * 16 + l15) * SA + lq;
        const float* b = Bs + buf * BK * SB + lq * SB + l15;
#pragma unroll
        for (int ks = KS0; ks < KS1; ++ks) {
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = a[i * 16 * SA + ks * 4];
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bv[j] = b[ks * 4 * SB + (Cfg::GW ? wn * CG + j * 16 : (j / TNG) * CG + (wn * TNG + j % TNG) * 16)];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };
    using K0 = std::integral_constant<int, 0>;
    using KH = std::integral_constant<int, S2VT_LAND_AT * (BK / 32)>;
    using K8 = std::integral_constant<int, BK / 4>;

    if constexpr (NBUF == 3) {
        // ---- fragment-pipelined loop (skinny-M kernels that run at one or two waves per SIMD): THREE LDS
        // stages and TWO fragment register sets, so that the MFMAs of chunk c read only registers whose
        // ds_reads were issued a whole iteration earlier, the ds_reads of chunk c+1 go out in one burst
        // right after the barrier, and chunk c+2 is stored to LDS under the MFMAs:
        //   global --(ring, PF chunks)--> registers --> LDS[(c+2)%3] --barrier--> fragments[(c+1)&1] --> MFMA(c)
        constexpr int KS = BK / 4;
        float fa[2][KS][TM], fb[2][KS][TN];
        auto read_frags = [&](int buf, float (&qa)[KS][TM], float (&qb)[KS][TN]) __attribute__((always_inline)) {
#ifdef S2VT_ABLATE
            if (g.dbg & 16) return;
#endif
            const float* a = As + buf * BM * SA + ((wm * TM) * 16 + l15) * SA + lq;
            const float* b = Bs + buf * BK * SB + lq * SB + l15;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int i = 0; i < TM; ++i) qa[ks][i] = a[i * 16 * SA + ks * 4];
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    qb[ks][j] = b[ks * 4 * SB + (Cfg::GW ? wn * CG + j * 16 : (j / TNG) * CG + (wn * TNG + j % TNG) * 16)];
            }
        };
        auto mfma_range = [&](float (&qa)[KS][TM], float (&qb)[KS][TN], auto ks0_, auto ks1_) __attribute__((always_inline)) {
            constexpr int KS0 = decltype(ks0_)::value, KS1 = decltype(ks1_)::value;
#ifdef S2VT_ABLATE
            if (g.dbg & 1) return;
#endif
#pragma unroll
            for (int ks = KS0; ks < KS1; ++ks)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[ks][i], qb[ks][j], acc[i][j], 0, 0, 0);
        };
        if (nchunks > 0) {
            // prologue: chunks 0, 1 -> LDS[0], LDS[1]; chunks 2 .. PF in flight (slot = chunk % PF); fragments of chunk 0
            S2VT_ISSUE(0, 0);
            S2VT_ISSUE(1, 1);
            if constexpr (VEC) wait_vmcnt<0>();
            land(0, ra[0], rb[0], pa[0], pb[0]);
            land(1, ra[1], rb[1], pa[1], pb[1]);
#pragma unroll
            for (int x = 2; x <= PF; ++x) S2VT_ISSUE(x, x % PF);
            __syncthreads();
            read_frags(0, fa[0], fb[0]);
            int c = 0, b1 = 1, b2 = 2;          // LDS stages of chunks c+1 and c+2
            bool more = true;
            while (more) {
#pragma unroll
                for (int j = 0; j < PF; ++j) {
                    if (more) {
                        read_frags(b1, fa[(j + 1) & 1], fb[(j + 1) & 1]);
                        S2VT_ISSUE(c + 1 + PF, (j + 1) % PF);
                        mfma_range(fa[j & 1], fb[j & 1], K0{}, KH{});
                        if constexpr (VEC) wait_vmcnt<WAITN>();
                        land(b2, ra[(j + 2) % PF], rb[(j + 2) % PF], pa[(j + 2) % PF], pb[(j + 2) % PF]);
                        mfma_range(fa[j & 1], fb[j & 1], KH{}, K8{});
#ifdef S2VT_ABLATE
                        if (!(g.dbg & 8))
#endif
                        __syncthreads();
                        ++c;
                        b1 = b2;                                   // stages advance to (c+1)%3, (c+2)%3
                        b2 = (b2 == 2) ? 0 : b2 + 1;
                        more = c < nchunks;
                    }
                }
            }
            drain_ring();
        }
    } else
    if constexpr (VEC) {
      if (nchunks > 0) {
        // ---- interleaved loop.  One wave per SIMD is the normal occupancy of the step kernels, so whatever is issued
        // between two barriers but not BETWEEN two MFMAs runs with the matrix pipe idle (s_memtime stamps of the
        // sequential form: load issue 600 + LDS stores 290 of 2600 clocks per chunk, gw80 tile).  Here the side
        // work of a chunk is cut into A4+B4 pieces and spliced after fixed MFMAs of the chunk's list (each
        // 16x16x4 fp32 MFMA leaves ~24 issue clocks free): first half = the buffer loads of chunk c+PF,
        // middle = the counted vmcnt wait, second half = LDS stores of chunk c+1; fragments are read one k-step
        // ahead in source order (an asm statement pins LDS reads, so nothing is left to the scheduler).
        constexpr int KS = BK / 4, MPK = TM * TN, NM = KS * MPK, HALF = NM / 2;
        const i32x4 rsW = make_rsrc(g.W);
        const i32x4 rsA0 = make_rsrc(sp0), rsA1 = make_rsrc(sp1), rsA2 = make_rsrc(sp2);
        const int akq = (tid % (BK / 4)) * 4;
        uint32_t avo0[A4], avo1[A4], avo2[A4], bvo[B4];
        int bkr[B4];
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            avo0[i] = aoff0[i] < 0 ? kOob : (uint32_t)(aoff0[i] + akq) * 4u;
            avo1[i] = aoff1[i] < 0 ? kOob : (uint32_t)(aoff1[i] + akq) * 4u;
            avo2[i] = aoff2[i] < 0 ? kOob : (uint32_t)(aoff2[i] + akq) * 4u;
        }
#pragma unroll
        for (int i = 0; i < B4; ++i) {
            const int idx = tid + i * NT;
            const int kr = idx / (BN / 4);
            const int col = (idx % (BN / 4)) * 4;
            const int grp = col / CG, cc = n0 + col % CG;
            const bool ok = (B4 * NT == BK * (BN / 4) || idx < BK * (BN / 4)) && cc < g.N;
            bvo[i] = ok ? (uint32_t)(kr * g.ldw + grp * g.gstride + cc) * 4u : kOob;
            bkr[i] = kr;
        }
        // The walk of the NEXT chunk to issue, kept as loop-carried scalars: segment index, k offset inside it and the
        // derived byte offsets advance by BK per chunk; only a segment change (rare, wave-uniform branch) re-selects
        // the descriptor and the per-lane row offsets.  Past the end of the walk krem_ = 0: every lane is out of
        // range and the ring loads return zeros without touching memory.
        // (macros at kernel scope on plain locals, not lambdas: see S2VT_ISSUE)
        int wseg_ = -1, koff_ = 0, krem_ = 0;
        i32x4 rsA_ = rsA0;
        uint32_t soffA_ = 0u, soffW_ = 0u;
        uint32_t avo_[A4];
#pragma unroll
        for (int i = 0; i < A4; ++i) avo_[i] = kOob;
#define S2VT_WALK_ENTER()                                                                                \
        do {                                                                                             \
            ++wseg_;                                                                                     \
            while (wseg_ < 3 && (wseg_ == 0 ? nch0 : (wseg_ == 1 ? nch1 : nch2)) == 0) ++wseg_;          \
            koff_ = 0;                                                                                   \
            if (wseg_ < 3) {                                                                             \
                rsA_ = wseg_ == 0 ? rsA0 : (wseg_ == 1 ? rsA1 : rsA2);                                   \
                krem_ = wseg_ == 0 ? slen0 : (wseg_ == 1 ? slen1 : slen2);                               \
                const int kw_ = (wseg_ == 0 ? skw0 : (wseg_ == 1 ? skw1 : skw2)) + kbeg;                 \
                soffA_ = 0u;                                                                             \
                soffW_ = (uint32_t)kw_ * (uint32_t)g.ldw * 4u;                                           \
                _Pragma("unroll") for (int i_ = 0; i_ < A4; ++i_)                                        \
                    avo_[i_] = wseg_ == 0 ? avo0[i_] : (wseg_ == 1 ? avo1[i_] : avo2[i_]);               \
            } else {                                                                                     \
                krem_ = 0;                                                                               \
            }                                                                                            \
        } while (0)
        // after the pieces of one chunk have been issued
#define S2VT_WALK_NEXT()                                                                                 \
        do {                                                                                             \
            krem_ -= BK;                                                                                 \
            soffA_ += (uint32_t)BK * 4u;                                                                 \
            soffW_ += (uint32_t)BK * (uint32_t)g.ldw * 4u;                                               \
            if (krem_ <= 0 && wseg_ < 3) S2VT_WALK_ENTER();                                              \
        } while (0)
#define S2VT_PIECE_ISSUE(P, SLOT)                                                                        \
        do {                                                                                             \
            if constexpr ((P) < A4) {                                                                    \
                bload16(ra[SLOT][(P) < A4 ? (P) : 0], akq < krem_ ? avo_[(P) < A4 ? (P) : 0] : kOob, rsA_, soffA_); \
            } else {                                                                                     \
                constexpr int i_ = (P) - A4 < B4 ? (P) - A4 : 0;                                         \
                bload16(rb[SLOT][i_], bkr[i_] < krem_ ? bvo[i_] : kOob, rsW, soffW_);                    \
            }                                                                                            \
        } while (0)
        auto land_piece = [&](int buf, auto p_, f32x4 (&qa)[A4], f32x4 (&qb)[B4]) __attribute__((always_inline)) {
            constexpr int P = decltype(p_)::value;
            if constexpr (P < A4) {
                pin(qa[P]);
                const f32x4 v = qa[P];
                const int idx = tid + P * NT;
                if (A4 * NT == BM * (BK / 4) || idx < BM * (BK / 4)) {
                    float* d = As + buf * BM * SA + (idx / (BK / 4)) * SA + (idx % (BK / 4)) * 4;
                    *reinterpret_cast<float2*>(d) = make_float2(v[0], v[1]);
                    *reinterpret_cast<float2*>(d + 2) = make_float2(v[2], v[3]);
                }
            } else {
                constexpr int i = P - A4;
                pin(qb[i]);
                const f32x4 v = qb[i];
                const int idx = tid + i * NT;
                if (B4 * NT == BK * (BN / 4) || idx < BK * (BN / 4))
                    *reinterpret_cast<f32x4*>(Bs + buf * BK * SB + (idx / (BN / 4)) * SB + (idx % (BN / 4)) * 4) = v;
            }
        };
        // MFMA after which piece p of each half is spliced
        auto splice = [](int p) constexpr { return ((2 * p + 1) * HALF) / (2 * LPC); };

        {   // prologue: chunk 0 -> LDS[0]; chunks 1 .. PF-1 in flight in ring slots 1 .. PF-1
            S2VT_WALK_ENTER();
            static_for<0, LPC>([&](auto p_) { constexpr int p = decltype(p_)::value; S2VT_PIECE_ISSUE(p, 0); });
            S2VT_WALK_NEXT();
            wait_vmcnt<0>();
            land(0, ra[0], rb[0], 0u, 0u);
#pragma unroll
            for (int j = 1; j < PF; ++j) {
                static_for<0, LPC>([&](auto p_) { constexpr int p = decltype(p_)::value; S2VT_PIECE_ISSUE(p, j); });
                S2VT_WALK_NEXT();
            }
            __syncthreads();
        }
        S2VT_STAMP_AT(0);
        int c = 0;
        bool more = true;
        while (more) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                if (more) {
                    const int buf = c & 1;
                    const float* a = As + buf * BM * SA + ((wm * TM) * 16 + l15) * SA + lq;
                    const float* b = Bs + buf * BK * SB + lq * SB + l15;
                    float av[2][TM], bv[2][TN];
                    auto read_frag = [&](auto ks_, float (&qa)[TM], float (&qb)[TN]) __attribute__((always_inline)) {
                        constexpr int ks = decltype(ks_)::value;
#pragma unroll
                        for (int i = 0; i < TM; ++i) qa[i] = a[i * 16 * SA + ks * 4];
#pragma unroll
                        for (int jj = 0; jj < TN; ++jj)
                            qb[jj] = b[ks * 4 * SB + (Cfg::GW ? wn * CG + jj * 16 : (jj / TNG) * CG + (wn * TNG + jj % TNG) * 16)];
                    };
                    read_frag(std::integral_constant<int, 0>{}, av[0], bv[0]);
                    static_for<0, NM>([&](auto n_) {
                        constexpr int n = decltype(n_)::value, ks = n / MPK, r = n % MPK, i = r / TN, jj = r % TN;
                        if constexpr (r == 0 && ks + 1 < KS) {
                            read_frag(std::integral_constant<int, ks + 1>{}, av[(ks + 1) & 1], bv[(ks + 1) & 1]);
                            __builtin_amdgcn_sched_barrier(0);     // hipcc otherwise sinks these reads to their first use
                        }
                        if constexpr (n == HALF) wait_vmcnt<WAITN>();
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks & 1][i], bv[ks & 1][jj], acc[i][jj], 0, 0, 0);
                        static_for<0, LPC>([&](auto p_) {
                            constexpr int p = decltype(p_)::value;
                            if constexpr (splice(p) == n) S2VT_PIECE_ISSUE(p, j);
                            if constexpr (HALF + splice(p) == n)
                                land_piece((c + 1) & 1, p_, ra[(j + 1) % PF], rb[(j + 1) % PF]);
                        });
                    });
                    S2VT_WALK_NEXT();
                    __syncthreads();
                    ++c;
                    more = c < nchunks;
                }
            }
        }
        drain_ring();
        S2VT_STAMP_AT(7);
#undef S2VT_WALK_ENTER
#undef S2VT_WALK_NEXT
#undef S2VT_PIECE_ISSUE
      }
    } else
    if (nchunks > 0) {
        // prologue: chunk 0 -> LDS[0]; chunks 1 .. PF-1 in flight in ring slots 1 .. PF-1
        S2VT_ISSUE(0, 0);
        if constexpr (VEC) wait_vmcnt<0>();
        land(0, ra[0], rb[0], pa[0], pb[0]);
#pragma unroll
        for (int j = 1; j < PF; ++j) S2VT_ISSUE(j, j);
        __syncthreads();
        S2VT_STAMP_AT(0);

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
                    S2VT_ISSUE(c + PF, j);
                    S2VT_STAMP_AT(1);
                    compute(c & 1, K0{}, KH{});
                    S2VT_STAMP_AT(2);
                    if constexpr (VEC) wait_vmcnt<WAITN>();
                    S2VT_STAMP_AT(3);
                    land((c + 1) & 1, ra[(j + 1) % PF], rb[(j + 1) % PF], pa[(j + 1) % PF], pb[(j + 1) % PF]);
                    S2VT_STAMP_AT(4);
                    compute(c & 1, KH{}, K8{});
                    S2VT_STAMP_AT(5);
#ifdef S2VT_ABLATE
                    if (!(g.dbg & 8))
#endif
                    __syncthreads();
                    S2VT_STAMP_AT(6);
                    ++c;
                    more = c < nchunks;
                }
            }
        }
        drain_ring();
        S2VT_STAMP_AT(7);
    }

#undef S2VT_ISSUE
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
    } else if constexpr (EPI == EPI_LSTM_GW) {
        // gate exchange: wave wn holds gate wn of the tile's units; z goes through LDS as Z[row][gate][unit]
        // (rows ZS floats apart), then every thread finishes (row, unit) pairs as in EPI_LSTM.
        constexpr int ZS = Cfg::ZS;
        const int H = g.N;
        __syncthreads();                       // every wave is done reading the operand buffers
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    smem[((wm * TM + i) * 16 + lq * 4 + r) * ZS + wn * CG + j * 16 + l15] = acc[i][j][r];
        __syncthreads();
        for (int pidx = tid; pidx < BM * CG; pidx += NT) {
            const int row = pidx / CG, uu = pidx % CG;
            const int m = m0 + row, u = n0 + uu;
            if (m >= g.M || u >= H) continue;
            const float* z = smem + row * ZS + uu;
            const float zi = z[0] + g.bias[u];
            const float zj = z[CG] + g.bias[H + u];
            const float zf = z[2 * CG] + g.bias[2 * H + u];
            const float zo = z[3 * CG] + g.bias[3 * H + u];
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
                    const float k01 = dropout_keep01(g.seed_lo, g.seed_hi, (uint32_t)g.video_id[m], (uint32_t)g.sample_id[m],
                                                     g.drop_code, (uint32_t)u, g.keep);
                    ov = (h / g.keep) * k01;
                }
                g.out[o] = ov;
            }
            if (g.gates) {
                float* gp = g.gates + (size_t)m * 4 * H + u;
                gp[0] = si; gp[H] = tj; gp[2 * H] = sf; gp[3 * H] = so;
            }
        }
    } else {  // EPI_PICK
        // Column tiles start at multiples of 16, so the quad l15 = 4q .. 4q+3 holds columns 4Q .. 4Q+3 of the
        // SAME four rows (lq*4 + r): quad lane e draws the Philox block of row r = e, the words are exchanged
        // with DPP (detmath.h).  All 64 lanes take part in every exchange (no divergence around it).
        const uint32_t e4 = (uint32_t)l15 & 3u;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mrow = m0 + (wm * TM + i) * 16 + lq * 4;
            int sid[4], vid[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool mok = mrow + r < g.M;
                sid[r] = mok ? g.sample_id[mrow + r] : -1;
                vid[r] = mok ? g.video_id[mrow + r] : 0;
            }
            const int sid_own = e4 == 0 ? sid[0] : e4 == 1 ? sid[1] : e4 == 2 ? sid[2] : sid[3];
            const int vid_own = e4 == 0 ? vid[0] : e4 == 1 ? vid[1] : e4 == 2 ? vid[2] : vid[3];
            float best[4];
            uint32_t bidx[4];
            bool have[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { best[r] = 0.0f; bidx[r] = 0xFFFFFFFFu; have[r] = false; }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + (wn * TN + j) * 16 + l15;
                u32x4 blk = {0u, 0u, 0u, 0u};
                if (sid_own >= 0)
                    blk = philox4x32_10((uint32_t)col >> 2, (uint32_t)vid_own, (uint32_t)sid_own, (uint32_t)g.step, g.seed_lo,
                                        g.seed_hi);
                uint32_t word[4];
                word[0] = quad_word_from<0>(blk, e4);
                word[1] = quad_word_from<1>(blk, e4);
                word[2] = quad_word_from<2>(blk, e4);
                word[3] = quad_word_from<3>(blk, e4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mrow + r;
                    if (m < g.M && col < g.N) {
                        float v = acc[i][j][r] + g.bias[col];
                        if (g.logits_out) g.logits_out[(size_t)m * g.ldc + col] = v;
                        if (sid[r] >= 0) v = v + gumbel_from_word(word[r]);
                        v = v + 0.0f;  // -0 -> +0 so that the integer order equals the float order
                        if (!have[r] || v > best[r]) { best[r] = v; bidx[r] = (uint32_t)col; have[r] = true; }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mrow + r;
                // reduce over the 16 lanes that hold this row's columns (ties -> lowest index)
                unsigned long long key = have[r] ? (((unsigned long long)orderable(best[r]) << 32) | (uint32_t)(~bidx[r])) : 0ull;
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    const unsigned long long o = __shfl_xor(key, off, 64);
                    key = o > key ? o : key;
                }
                if (l15 == 0 && m < g.M && key != 0ull) atomicMax(&g.pick[m], key);
            }
        }
    }
#ifdef S2VT_STAMP
    S2VT_STAMP_AT(8);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) atomicAdd(&s2vt_stamp_acc[i], st_acc[i]);
        atomicAdd(&s2vt_stamp_acc[9], 1ull);
        atomicAdd(&s2vt_stamp_acc[10], (unsigned long long)nchunks);
    }
#endif
}

}  // namespace s2vt
