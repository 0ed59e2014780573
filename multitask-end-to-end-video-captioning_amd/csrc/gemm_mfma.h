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
// * Tiling: 64-lane waves, WM x WN MFMA waves per workgroup (+ PW loader waves in the step tiles), TM x TN 16x16
//   accumulators per wave, BK = 32 (64) K-chunk, asm-issued raw-buffer loads -> register ring -> LDS double buffer,
//   one barrier per chunk, the loads and LDS stores spliced between the MFMAs.  LDS images: A as four k%4 planes
//   with a row-XOR swizzle (one conflict-free ds_read_b128 feeds four k-steps), B rows == 16 (mod 32) floats apart.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "detmath.h"

namespace s2vt {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;

enum { EPI_STORE = 0, EPI_LSTM = 1, EPI_PICK = 2, EPI_LSTM_GW = 3 };
constexpr int EPI_STORE_NT = 4;     // launcher-side id only: EPI_STORE with the W operand given as W^T ([N][K], K contiguous)

struct ASeg {
    const float* ptr;    // [rows, ld] row-major; nullptr = segment absent
    const int* rowidx;   // optional gather: row(m) = rowidx[m]
    const unsigned long long* rowkey;  // optional gather through packed PICK results: row(m) = ~low32(rowkey[m])
    int ld;
    int k;               // segment length along K
    int kw;              // first row of W this segment multiplies
    int rowmod;          // >0: row(m) = m % rowmod (applied before rowidx)
    int rowkey_stride;   // 64-bit words between the rowkey entries of consecutive rows (0 = 1, dense)
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
    int xcd_map;         // 1: XCD-aware tile order (set by the launcher for skinny-M shapes); 2: L2-sized supertiles (many-tile shapes)
    int sup_gm, sup_gn;  //    xcd_map == 2: row tiles x column tiles of a supertile (~ the workgroups one XCD runs at a time)
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
    int pick_stride;            // 64-bit words between consecutive rows' entries (0 = 1, dense).  The sampler spreads them one
                                // per 128-byte line (kPickStride): every column tile of the launch does an atomicMax per row, and
                                // 16 rows per line made ~4000 serialized atomics per line at M = 64 (15-18 us of a 48 us launch)
    float* logits_out;          // optional [M, ldc]
    // live-row launches (the OM instantiations: sampler steps that skip finished samples).  Row m of the launch is a COMPACT index;
    // every per-row access -- A segments (before their own rowmod / rowidx / rowkey), cinit, c_prev, the state / pick outputs, the
    // noise ids -- uses row omap[m] of the caller's arrays, and only the first *m_dev rows exist (tiles behind them return at once).
    const int* omap;
    const int* m_dev;
};

// ---- loads the compiler does not schedule (cdna_hip_programming.md §5.7): hipcc sinks ordinary prefetch loads next
// to their first use and drains them with vmcnt(0); issued as asm they stay where they are written, and the ring is
// drained with a hand-counted s_waitcnt.
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

template <int WM, int WN, int TM, int TN, int NG, int EPI, bool VEC, int BKT = 32, int PW = 0, bool BT = false, int DM = 0>
struct GemmCfg {
    // K-chunk depth of this configuration (k per barrier): 32 by default; 64 for the tiles whose chunk holds few
    // MFMAs per wave (M = 64 step kernels: 8 per chunk at 32), where the per-chunk barrier and waits dominate
    static constexpr int BK = BKT;
    static constexpr int KQ = BKT / 4;                        // k-steps (MFMAs along k) per chunk
    // PW > 0: PW extra LOADER waves per workgroup (loads + LDS stores only) beside the WM*WN MFMA waves (fragment reads
    // + MFMAs only); PW == 0: every wave does both, interleaved.
    static constexpr int NTC = 64 * WM * WN;                  // MFMA threads
    static constexpr int NTL = PW > 0 ? 64 * PW : NTC;        // loader threads
    static constexpr int NT = NTC + 64 * PW;                  // workgroup size
    static constexpr int BM = WM * TM * 16;
    static constexpr int BN = WN * TN * 16;
    // A image: four planes (one per k % 4) of BM rows of KQ floats, 16-byte groups XOR-swizzled by row (see the
    // kernel); B image: [BK][SB]
    static constexpr int RS = KQ;
    static constexpr int NG4 = KQ / 4;                        // 16-byte groups per plane row (2 or 4)
    static constexpr int SWZ_SHIFT = NG4 == 2 ? 3 : 2;        // rows that share a bank group differ in bit(s) >= this
    static constexpr int PL = BM * RS;
    static constexpr int ABUF = 4 * PL;
    // EPI_LSTM_GW ("gate per wave"): the four waves along N each take ONE gate column group of the same
    // TN*16 hidden units (WN == 4 == NG), so a workgroup is BM rows x TN*16 units and the grid can be cut
    // to ~one workgroup per CU for any M; the gates of a unit meet through LDS in the epilogue.
    static constexpr bool GW = (EPI == EPI_LSTM_GW);
    static constexpr int TNG = GW ? TN : TN / NG;            // subtiles per group per wave
    static constexpr int CG = GW ? TN * 16 : WN * TNG * 16;  // tile columns per group
    static constexpr int ZS = 4 * CG + 4;                    // GW gate-exchange image: floats per row
    static constexpr int SB = (BN % 32 == 16) ? BN : BN + 16;
    static constexpr int A4 = (BM * KQ + NTL - 1) / NTL;       // float4 per loader thread per chunk
    // BT: the W operand is stored transposed, W^T[n][k] with k contiguous (the backward data-gradient products
    // dY . W^T read the forward weight matrix as it lies): it is loaded, stored and read exactly like A -- float4s
    // along k, four k % 4 planes [n][k / 4] with the row swizzle, one ds_read_b128 per four k-steps
    static constexpr int BITEMS = BT ? BN * KQ : BK * (BN / 4);   // float4 items of one B chunk
    static constexpr int B4 = (BITEMS + NTL - 1) / NTL;
    static constexpr int BBUF = BT ? 4 * BN * KQ : BK * SB;       // floats per B stage
    // DM > 0 (with PW > 0): the PW loader waves feed a DM-stage LDS ring by LDS-DMA (buffer_load_dwordx4 ... lds) -- no staging
    // registers, no ds_write pass -- and the images are laid out for that: A rows as they lie in memory ([BM][BK], 16-byte groups
    // XOR-swizzled through the DMA source address), B rows unpadded ([BK][BN], rotated by 16 floats per k % 4); see the kernel.
    static constexpr int DSTAGE = BM * BKT + BKT * BN;
    static constexpr int LOOP_FLOATS = DM > 0 ? DM * DSTAGE : 2 * (ABUF + BBUF);
    static constexpr int LDS_FLOATS = (GW && BM * ZS > LOOP_FLOATS) ? BM * ZS : LOOP_FLOATS;
    // Prefetch ring depth (chunks in flight per thread), from a register budget: small tiles (few accumulators) get a
    // deeper ring, big tiles run two workgroups per CU and keep two chunks.
#ifndef S2VT_PF_BUDGET
#define S2VT_PF_BUDGET 48
#endif
    // B fragments ([k][n] image, one ds_read_b32 per k-step and column subtile) are read PD k-steps ahead of the MFMAs that
    // consume them.  One k-step ahead is enough when a k-step holds >= 4 MFMAs (>= 128 clocks against ~64-100 of LDS
    // latency); the thin step tiles (gw16 / gw32 / gw48: 1-3 MFMAs per k-step, ONE dependent accumulator chain at M <= 64)
    // waited for every fragment -- measured ~78 clocks per MFMA instead of the chain's 40 -- so they read 4 / 2 ahead.
    static constexpr int MPK_ = TM * TN;
    static constexpr int PD = MPK_ >= 4 ? 1 : (MPK_ >= 2 ? 2 : 4);
    static constexpr int PF_RAW = (PW > 0 ? 2 : 1) * S2VT_PF_BUDGET / (4 * (A4 + B4));   // loader waves hold no accumulators
    static constexpr int PF = PF_RAW < 2 ? 2 : (PF_RAW > 6 ? 6 : PF_RAW);
    static_assert(NG4 == 2 || NG4 == 4, "A planes: two or four b128 groups per row");
    static_assert(!BT || (EPI == EPI_STORE && NG == 1 && (PW == 0 || DM > 0)), "transposed-W form: plain store tiles only");
    static_assert(GW || TN % NG == 0, "TN must split evenly over the column groups");
    static_assert(!GW || (WN == 4 && NG == 4), "gate-per-wave needs four waves along N");
    static_assert(DM == 0 || (PW > 0 && VEC), "LDS-DMA ring: loader waves, aligned operands");
};

template <int WM, int WN, int TM, int TN, int NG, int EPI, bool VEC, int BKT = 32, int PW = 0, bool BT = false, bool OM = false, int DM = 0>
__global__ __launch_bounds__(64 * (WM * WN + PW)) void gemm_kernel(const GemmArgs g)
{
    using Cfg = GemmCfg<WM, WN, TM, TN, NG, EPI, VEC, BKT, PW, BT, DM>;
    constexpr int BK = Cfg::BK, KQ = Cfg::KQ, RS = Cfg::RS, PL = Cfg::PL, ABUF = Cfg::ABUF;   // (BK shadows the namespace-scope default)
    constexpr int NG4 = Cfg::NG4, SWZ_SHIFT = Cfg::SWZ_SHIFT;
    constexpr int NT = Cfg::NT, NTL = Cfg::NTL, NCW = WM * WN, BM = Cfg::BM, BN = Cfg::BN, TNG = Cfg::TNG, CG = Cfg::CG, SB = Cfg::SB;
    constexpr int A4 = Cfg::A4, B4 = Cfg::B4, BITEMS = Cfg::BITEMS, BBUF = Cfg::BBUF, PLB = BN * KQ;
    constexpr int PF = Cfg::PF;

#ifdef S2VT_STAMP
    unsigned long long st_acc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = stamp_now();
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][4][BM][RS]
    float* Bs = smem + 2 * ABUF;            // [2][BK][SB]  (BT: [2][4][BN][KQ] planes, as A)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool mma_wave = PW == 0 || wave < NCW;          // this wave owns accumulators
    const int ltid = PW > 0 ? tid - Cfg::NTC : tid;       // index among the loader threads (negative: not a loader)
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
    if (g.xcd_map == 2) {
        // Many-tile shapes.  The tile grid is walked supertile by supertile (sup_gm row tiles x sup_gn column tiles, row
        // tiles fastest inside one); XCD b % 8 takes a contiguous, equally long stretch of that walk, so the ~64-96
        // workgroups an XCD runs at a time form one supertile: they walk K together, every A chunk is fetched into that
        // XCD's L2 once for sup_gn column tiles and every W chunk once for sup_gm row tiles (the previous order ran ONE row
        // tile x 64 column tiles at a time: the whole weight matrix re-streamed per row tile -- 2.5 GB of fetches for the
        // 0.38 GB logits product).  Bijective for any tile count; edge supertiles are simply smaller.
        const int nwg = ntile_m * ntile_n, q = nwg >> 3, r = nwg & 7;
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        const int p = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
        const int rowblk = g.sup_gm * ntile_n;
        const int sm = p / rowblk, p1 = p - sm * rowblk;
        const int hm = min(g.sup_gm, ntile_m - sm * g.sup_gm);
        const int colblk = hm * g.sup_gn;
        const int sn = p1 / colblk, within = p1 - sn * colblk;
        tile_m = sm * g.sup_gm + within % hm;
        tile_n = sn * g.sup_gn + within / hm;
    } else if (g.xcd_map) {
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
    // rows of this launch: g.M, or (OM) the device-resident live count -- a tile behind it has nothing to do
    int MM = g.M;
    if constexpr (OM) {
        if (g.m_dev) { const int md = *g.m_dev; MM = md < MM ? md : MM; }
        if (m0 >= MM) return;
    }
    const bool omapped = OM && g.omap != nullptr;
    auto orow = [&](int m) __attribute__((always_inline)) { if constexpr (OM) return g.omap ? g.omap[m] : m; else return m; };

    f32x4 acc[TM][TN];
    // initial accumulator: +0 or a carried partial chain
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (g.cinit && mma_wave) {
                const int cu = Cfg::GW ? n0 + j * 16 + l15 : n0 + (wn * TNG + j % TNG) * 16 + l15;   // column within its group
                const int col = (Cfg::GW ? wn : j / TNG) * g.gstride + cu;
                const bool cok = cu < g.N;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int m = m0 + (wm * TM + i) * 16 + lq * 4 + r;
                    if (m < MM && cok) {
                        m = orow(m);
                        if (g.cinit_rowmod > 0) m %= g.cinit_rowmod;
                        v[r] = g.cinit[(size_t)m * g.ldcinit + col];
                    }
                }
            }
            acc[i][j] = v;
        }

    // Epilogue operands that do not depend on the contraction (bias, noise-stream ids) are loaded HERE, under the first
    // chunk's latency, instead of as a chain of dependent loads in the epilogue; hipcc sinks any plain load to its
    // first use unless the value is pinned in a register (below).  (Worth ~1 % of the PICK launch.)
    float ep_bias[TN];
    int ep_sid[TM][4], ep_vid[TM][4];
    if constexpr (EPI == EPI_STORE || EPI == EPI_PICK) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cc = n0 + (wn * TNG + j % TNG) * 16 + l15;
            const int col = EPI == EPI_PICK ? n0 + (wn * TN + j) * 16 + l15 : (j / TNG) * g.gstride + cc;
            ep_bias[j] = (g.bias && mma_wave && (EPI == EPI_PICK ? col : cc) < g.N) ? g.bias[col] : 0.0f;
        }
    }
    if constexpr (EPI == EPI_PICK) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + (wm * TM + i) * 16 + lq * 4 + r;
                const bool mok = mma_wave && m < MM;
                ep_sid[i][r] = mok ? g.sample_id[orow(m)] : -1;
                ep_vid[i][r] = mok ? g.video_id[orow(m)] : 0;
            }
    }
    auto pin_epilogue_operands = [&]() __attribute__((always_inline)) {
        if constexpr (EPI == EPI_STORE || EPI == EPI_PICK) {
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(ep_bias[j]));
        }
        if constexpr (EPI == EPI_PICK) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    asm volatile("" : "+v"(ep_sid[i][r]));
                    asm volatile("" : "+v"(ep_vid[i][r]));
                }
        }
    };

    // per-thread staging ring: PF chunks in flight between HBM/L2 and the LDS double buffer
    f32x4 ra[PF][A4], rb[PF][B4];
    constexpr int LPC = A4 + B4;                                   // asm-issued loads per chunk per thread
    constexpr int WAITN = ((PF - 1) * LPC > 63) ? 63 : (PF - 1) * LPC;

    // ---- wave-uniform description of the K walk: up to three A segments, each a whole number of chunks
    const int kbeg = g.splits > 1 ? (int)blockIdx.y * g.kper : 0;
    auto seg_len = [&](const float* ptr, int sk, int i) __attribute__((always_inline)) {
        int k = 0;
        if (i < g.nseg && ptr != nullptr) {
            k = sk - kbeg;
            if (g.splits > 1 && k > g.kper) k = g.kper;
            if (k < 0) k = 0;
        }
        return k;
    };
    // The three segment descriptors as named scalars: a run-time index into the by-value argument struct makes hipcc
    // copy the struct to scratch, and scratch loads share vmcnt with the ring (measured: every chunk then drains the
    // whole prefetch ring, 2.2 us per chunk).  uniform() makes each field an opaque SGPR value: left alone, hipcc turns
    // "s == 0 ? seg[0].f : ..." into a load through a selected ADDRESS, with the same effect.  For the same reason
    // every run-time selection between the per-segment locals below is written at KERNEL scope (macros), never
    // inside a lambda: between by-reference captures it becomes a load at a run-time offset into the closure.
    // A plain segment (no broadcast / gather) is addressed from the first row of THIS tile: the 32-bit offsets of the
    // vector path then span BM rows, whatever the size of the matrix (a 71680 x 9972 logits gradient is 2.9 GB)
    auto seg_base = [&](int i) __attribute__((always_inline)) -> const float* {
        const ASeg& sg = g.seg[i];
        const bool plain = sg.ptr && sg.rowmod <= 0 && !sg.rowidx && !sg.rowkey && !omapped;
        return uniform(plain ? sg.ptr + (size_t)m0 * sg.ld : sg.ptr);
    };
    const float* const sp0 = seg_base(0);
    const float* const sp1 = seg_base(1);
    const float* const sp2 = seg_base(2);
    const int skw0 = uniform(g.seg[0].kw), skw1 = uniform(g.seg[1].kw), skw2 = uniform(g.seg[2].kw);
    const int slen0 = uniform(seg_len(sp0, g.seg[0].k, 0)), slen1 = uniform(seg_len(sp1, g.seg[1].k, 1)),
              slen2 = uniform(seg_len(sp2, g.seg[2].k, 2));
    const int nch0 = (slen0 + BK - 1) / BK, nch1 = (slen1 + BK - 1) / BK, nch2 = (slen2 + BK - 1) / BK;
    const int nchunks = nch0 + nch1 + nch2;

    // Row offsets of this thread's A slots for every segment, resolved ONCE (gather / broadcast index
    // loads happen here, never inside the pipelined loop).  -1 marks a row beyond M / an absent segment.
    int aoff0[A4], aoff1[A4], aoff2[A4];
    auto row_offsets = [&](int sl, int rowmod, const int* rowidx, const unsigned long long* rowkey, int rks, int ld, int (&ao)[A4]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            const int idx = ltid + i * NTL;
            int m = m0 + idx / KQ;
            int off = -1;
            if (sl > 0 && idx >= 0 && idx < BM * KQ && m < MM) {
                if (rowmod <= 0 && !rowidx && !rowkey && !omapped) m -= m0;          // plain segment: relative to the tile's first row (seg_base)
                if (omapped) m = orow(m);
                if (rowmod > 0) m %= rowmod;
                if (rowidx) m = rowidx[m];
                if (rowkey) m = (int)(~(uint32_t)rowkey[(size_t)m * (rks > 0 ? rks : 1)]);
                off = m * ld + kbeg;
            }
            ao[i] = off;
        }
    };
    row_offsets(slen0, g.seg[0].rowmod, g.seg[0].rowidx, g.seg[0].rowkey, g.seg[0].rowkey_stride, g.seg[0].ld, aoff0);
    row_offsets(slen1, g.seg[1].rowmod, g.seg[1].rowidx, g.seg[1].rowkey, g.seg[1].rowkey_stride, g.seg[1].ld, aoff1);
    row_offsets(slen2, g.seg[2].rowmod, g.seg[2].rowidx, g.seg[2].rowkey, g.seg[2].rowkey_stride, g.seg[2].ld, aoff2);

    // ---- LDS images.  A is stored as four PLANES, one per lq = k % 4: As[stage][lq][row][k / 4].  The MFMA lane
    // (l15, lq) consumes A[row l15][k = 4*ks + lq] at k-step ks, so its operands for FOUR consecutive k-steps are 16
    // contiguous bytes of plane lq: one ds_read_b128 instead of four ds_read_b32.  ds_read_b128 is serviced in 16-lane
    // groups that pair the l15 ranges of two lq values ({0-3,12-15} of one with {4-11} of the next): conflict-free iff
    // the 16 rows hit 16 distinct 4-bank groups whatever lq -- planes are a multiple of 64 dwords apart (BM % 16 == 0)
    // and, rows being only KQ = 8 (16) floats long, the 16-byte group g of row r is stored at g ^ swz(r), swz = the
    // row bits that would otherwise alias (r >> 3 & 1 for two groups per row, r >> 2 & 3 for four).  No padding.
    // B stays [k][n] (rows SB == 16 mod 32 floats: conflict-free ds_read_b32).  A float4 of the ring (4 consecutive k
    // of one row) lands as one dword in each plane (32 consecutive lanes cover 32 distinct banks).
    auto a_store = [&](int buf, int idx, const f32x4& v) __attribute__((always_inline)) {
        const int r = idx / KQ, kq = idx % KQ;
        float* d = As + buf * ABUF + r * RS + (((kq >> 2) ^ ((r >> SWZ_SHIFT) & (NG4 - 1))) << 2) + (kq & 3);
        d[0] = v[0];
        d[PL] = v[1];
        d[2 * PL] = v[2];
        d[3 * PL] = v[3];
    };
    auto b_store = [&](int buf, int idx, const f32x4& v) __attribute__((always_inline)) {
        if constexpr (BT) {
            const int r = idx / KQ, kq = idx % KQ;
            float* d = Bs + buf * BBUF + r * KQ + (((kq >> 2) ^ ((r >> SWZ_SHIFT) & (NG4 - 1))) << 2) + (kq & 3);
            d[0] = v[0];
            d[PLB] = v[1];
            d[2 * PLB] = v[2];
            d[3 * PLB] = v[3];
        } else {
            *reinterpret_cast<f32x4*>(Bs + buf * BBUF + (idx / (BN / 4)) * SB + (idx % (BN / 4)) * 4) = v;
        }
    };
    // this lane's fragment bases inside a stage
    const int a_frag = lq * PL + ((wm * TM) * 16 + l15) * RS;
    const int a_swz = (l15 >> SWZ_SHIFT) & (NG4 - 1);    // swz(row) of every fragment row of this lane (rows differ by multiples of 16)
    const int b_frag = BT ? lq * PLB + l15 * KQ : lq * SB + l15;
    auto b_col = [&](int j) __attribute__((always_inline)) { return Cfg::GW ? wn * CG + j * 16 : (j / TNG) * CG + (wn * TNG + j % TNG) * 16; };

    if constexpr (DM > 0) {
      // ---- LDS-DMA ring (round 6).  tools/micro/stream_probe.hip: a wave moves ~14.5 GB/s through buffer_load_dwordx4 however many
      // loads it keeps in flight (one 1-KiB wave-instruction per ~170 clocks), a CU up to ~100 GB/s from L2 given enough waves -- and
      // every sampler-step kernel of round 5 sat at ~20 GB/s per CU, its loads spliced into the MFMA stream (each issue holds the wave
      // for that long) or carried by loader waves that also ran the LDS stores.  Here the PW loader waves do NOTHING but issue DMA
      // pieces (global -> LDS, 1 KiB per wave-instruction, no staging registers, no ds_write), DM stages deep, and the MFMA waves
      // nothing but fragment reads and MFMAs.  Same products in the same order per accumulator as the register-staged loop: the
      // images differ, the arithmetic does not.
      //   A stage: [BM][BK] floats, row r as it lies in memory, its 16-byte group q stored at position q ^ f(r) (the DMA source offset
      //            does the permutation: a lane's 16 destination bytes are fixed, its source is free).  Fragment = ds_read_b32 of
      //            float (k & 3) of group (k >> 2) ^ f(r): with f(r) = (r >> 1) & 7 for 128-byte rows (r & 15 for 256-byte rows) the
      //            16 rows x 4 k of a wave hit 64 distinct banks.
      //   B stage: [BK][BN] floats, unpadded; row kk rotated by R(kk) floats (16 (kk & 3) when rows are a multiple of 64 floats,
      //            16 (kk >> 1 & 1) when they are 32 mod 64): the four k of a fragment read land in four disjoint 16-bank ranges.
      static_assert(BK == 32 || BK == 64, "LDS-DMA ring: 128- or 256-byte A rows");
      constexpr int NGR = BK / 4, RPP = 256 / BK;                   // 16-byte groups per A row; A rows per 1-KiB piece
      constexpr int AST = BM * BK, BST = BK * BN, STG = AST + BST;
      constexpr int APC = BM / RPP, BPC = BST / 256;                // 1-KiB pieces per chunk
      static_assert(BM % RPP == 0 && BST % 256 == 0 && APC % PW == 0 && BPC % PW == 0, "pieces divide evenly over the loader waves");
      static_assert(BT || BN % 64 == 0 || BN % 64 == 32, "B rotation");
      //   B stage, W given transposed (BT: W^T[n][k], k contiguous -- the backward data-gradient products): [BN][BK], rows of n laid out
      //            and read exactly like A's rows.
      constexpr int APW = APC / PW, BPW = BPC / PW, PPW = APW + BPW;
      constexpr int WAITD = (DM - 2) * PPW;
      static_assert(DM >= 2 && WAITD <= 63, "vmcnt range");
      typedef __attribute__((address_space(3))) void* lds_ptr;
      auto brot = [](int kk) constexpr { return (BN % 64 == 0) ? 16 * (kk & 3) : 16 * ((kk >> 1) & 1); };
      auto fswz = [](int r) constexpr { return BK == 32 ? (r >> 1) & 7 : r & 15; };
      if (nchunks > 0) {
        if (!mma_wave) {
          // ================= loader waves
          __builtin_amdgcn_s_setprio(3);
          const int lw = wave - NCW;
          uint32_t da0[APW], da1[APW], da2[APW], db[BPW];
          int dq4[APW], dkk[BPW];
          auto row_off1 = [&](int r, int sl, int rowmod, const int* rowidx, const unsigned long long* rowkey, int rks, int ld) __attribute__((always_inline)) {
              int m = m0 + r;
              int off = -1;
              if (sl > 0 && m < MM) {
                  if (rowmod <= 0 && !rowidx && !rowkey && !omapped) m -= m0;
                  if (omapped) m = orow(m);
                  if (rowmod > 0) m %= rowmod;
                  if (rowidx) m = rowidx[m];
                  if (rowkey) m = (int)(~(uint32_t)rowkey[(size_t)m * (rks > 0 ? rks : 1)]);
                  off = m * ld + kbeg;
              }
              return off;
          };
#pragma unroll
          for (int s_ = 0; s_ < APW; ++s_) {
              const int r = (s_ * PW + lw) * RPP + lane / NGR, q = (lane % NGR) ^ fswz(r);
              dq4[s_] = 4 * q;
              const int o0 = row_off1(r, slen0, g.seg[0].rowmod, g.seg[0].rowidx, g.seg[0].rowkey, g.seg[0].rowkey_stride, g.seg[0].ld);
              const int o1 = row_off1(r, slen1, g.seg[1].rowmod, g.seg[1].rowidx, g.seg[1].rowkey, g.seg[1].rowkey_stride, g.seg[1].ld);
              const int o2 = row_off1(r, slen2, g.seg[2].rowmod, g.seg[2].rowidx, g.seg[2].rowkey, g.seg[2].rowkey_stride, g.seg[2].ld);
              da0[s_] = o0 < 0 ? kOob : (uint32_t)(o0 + 4 * q) * 4u;
              da1[s_] = o1 < 0 ? kOob : (uint32_t)(o1 + 4 * q) * 4u;
              da2[s_] = o2 < 0 ? kOob : (uint32_t)(o2 + 4 * q) * 4u;
          }
#pragma unroll
          for (int s_ = 0; s_ < BPW; ++s_) {
              if constexpr (BT) {
                  const int r = (s_ * PW + lw) * RPP + lane / NGR, q = (lane % NGR) ^ fswz(r);    // row of W^T = output column n0 + r
                  dkk[s_] = 4 * q;
                  db[s_] = n0 + r < g.N ? (uint32_t)((n0 + r) * g.ldw + 4 * q) * 4u : kOob;
              } else {
                  const int L = (s_ * PW + lw) * 256 + lane * 4;         // float index inside the B stage this lane's 16 bytes land at
                  const int kk = L / BN, pp = L % BN;
                  const int c0 = (pp - brot(kk) + BN) % BN;              // tile column of the group stored there
                  const int grp = c0 / CG, cc = n0 + c0 % CG;
                  dkk[s_] = kk;
                  db[s_] = cc < g.N ? (uint32_t)(kk * g.ldw + grp * g.gstride + cc) * 4u : kOob;
              }
          }
          const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.W), 0, (int)kOob, 0x00020000);
          const __amdgpu_buffer_rsrc_t rA0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sp0), 0, (int)kOob, 0x00020000);
          const __amdgpu_buffer_rsrc_t rA1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sp1), 0, (int)kOob, 0x00020000);
          const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sp2), 0, (int)kOob, 0x00020000);
          // the walk of the next chunk to issue (segment, k left in it, byte advances) -- the scalars of the register-staged loop
          int wseg_ = -1, krem_ = 0;
          uint32_t soffA_ = 0u, soffW_ = 0u;
          uint32_t dva_[APW];
#pragma unroll
          for (int i_ = 0; i_ < APW; ++i_) dva_[i_] = kOob;
#define S2VT_DWALK_ENTER()                                                                               \
          do {                                                                                           \
              ++wseg_;                                                                                   \
              while (wseg_ < 3 && (wseg_ == 0 ? nch0 : (wseg_ == 1 ? nch1 : nch2)) == 0) ++wseg_;        \
              if (wseg_ < 3) {                                                                           \
                  krem_ = wseg_ == 0 ? slen0 : (wseg_ == 1 ? slen1 : slen2);                             \
                  const int kw_ = (wseg_ == 0 ? skw0 : (wseg_ == 1 ? skw1 : skw2)) + kbeg;               \
                  soffA_ = 0u;                                                                           \
                  soffW_ = (uint32_t)kw_ * (BT ? 1u : (uint32_t)g.ldw) * 4u;                             \
                  _Pragma("unroll") for (int i_ = 0; i_ < APW; ++i_)                                     \
                      dva_[i_] = wseg_ == 0 ? da0[i_] : (wseg_ == 1 ? da1[i_] : da2[i_]);                \
              } else {                                                                                   \
                  krem_ = 0;                                                                             \
              }                                                                                          \
          } while (0)
#define S2VT_DWALK_NEXT()                                                                                \
          do {                                                                                           \
              krem_ -= BK;                                                                               \
              soffA_ += (uint32_t)BK * 4u;                                                               \
              soffW_ += (uint32_t)BK * (BT ? 1u : (uint32_t)g.ldw) * 4u;                                 \
              if (krem_ <= 0 && wseg_ < 3) S2VT_DWALK_ENTER();                                           \
          } while (0)
          auto issue_chunk = [&](int stage) __attribute__((always_inline)) {
              float* const sb = smem + stage * STG;
              // (the descriptor of the segment being walked: a wave-uniform choice between three SGPR quads)
              static_for<0, APW>([&](auto s_) {
                  constexpr int sl = decltype(s_)::value;
                  const uint32_t vo = dq4[sl] < krem_ ? dva_[sl] : kOob;
                  float* const dst = sb + (sl * PW + lw) * 256;
                  if (wseg_ <= 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA0, (lds_ptr)dst, 16, vo, soffA_, 0, 0);
                  else if (wseg_ == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA1, (lds_ptr)dst, 16, vo, soffA_, 0, 0);
                  else __builtin_amdgcn_raw_ptr_buffer_load_lds(rA2, (lds_ptr)dst, 16, vo, soffA_, 0, 0);
              });
              static_for<0, BPW>([&](auto s_) {
                  constexpr int sl = decltype(s_)::value;
                  const uint32_t vo = dkk[sl] < krem_ ? db[sl] : kOob;
                  __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr)(sb + AST + (sl * PW + lw) * 256), 16, vo, soffW_, 0, 0);
              });
          };
          S2VT_DWALK_ENTER();
#pragma unroll
          for (int j = 0; j < DM - 1; ++j) {
              issue_chunk(j);
              S2VT_DWALK_NEXT();
          }
          wait_vmcnt<WAITD>();
          __syncthreads();
          int stage_n = DM - 1;                                      // the stage chunk c + DM - 1 goes to
          for (int c = 0; c < nchunks; ++c) {
              issue_chunk(stage_n);
              S2VT_DWALK_NEXT();
              stage_n = stage_n + 1 == DM ? 0 : stage_n + 1;
              S2VT_STAMP_AT(2);                                      // (dev build) loader: issue
              wait_vmcnt<WAITD>();                                   // chunk c + 1 has landed; the younger ones stay in flight
              S2VT_STAMP_AT(3);                                      // loader: ring wait
              __syncthreads();
              S2VT_STAMP_AT(7);                                      // loader: barrier
          }
          wait_vmcnt<0>();
#undef S2VT_DWALK_ENTER
#undef S2VT_DWALK_NEXT
        } else {
          // ================= MFMA waves: fragment reads + MFMAs, nothing else
          const int fl4 = fswz(l15) << 2;                             // (rows of this lane differ by multiples of 16: one swizzle)
          int aofs[KQ], bofs[TN];
#pragma unroll
          for (int ks = 0; ks < KQ; ++ks) aofs[ks] = ((wm * TM) * 16 + l15) * BK + ((ks * 4) ^ fl4) + lq;
#pragma unroll
          for (int j = 0; j < TN; ++j)
              bofs[j] = BT ? AST - ((wm * TM) * 16 + l15) * BK + (b_col(j) + l15) * BK        // (+ aofs[ks]: the same swizzled k offset, row b_col(j) + l15 of the W^T image)
                           : AST + lq * BN + (b_col(j) + l15 + brot(lq)) % BN;
          pin_epilogue_operands();
          __syncthreads();
          S2VT_STAMP_AT(0);
          int stage_c = 0;
          for (int c = 0; c < nchunks; ++c) {
              const float* st = smem + stage_c * STG;
              constexpr int MPK = TM * TN, PD = Cfg::PD;          // fragments are read PD k-steps ahead of their MFMAs (thin tiles: further)
              float av[PD + 1][TM], bw[PD + 1][TN];
              auto read_k = [&](auto ks_, float (&qa)[TM], float (&qb)[TN]) __attribute__((always_inline)) {
                  constexpr int ks = decltype(ks_)::value;
#pragma unroll
                  for (int i = 0; i < TM; ++i) qa[i] = st[aofs[ks] + i * 16 * BK];
#pragma unroll
                  for (int jj = 0; jj < TN; ++jj) qb[jj] = BT ? st[bofs[jj] + aofs[ks]] : st[bofs[jj] + ks * 4 * BN];
              };
              static_for<0, (PD < KQ ? PD : KQ)>([&](auto k_) { constexpr int k = decltype(k_)::value; read_k(k_, av[k % (PD + 1)], bw[k % (PD + 1)]); });
              S2VT_STAMP_AT(1);
              static_for<0, KQ * MPK>([&](auto n_) {
                  constexpr int n = decltype(n_)::value, ks = n / MPK, r = n % MPK, i = r / TN, jj = r % TN;
                  if constexpr (r == 0) {
                      if constexpr (ks + PD < KQ)
                          read_k(std::integral_constant<int, ks + PD>{}, av[(ks + PD) % (PD + 1)], bw[(ks + PD) % (PD + 1)]);
                      if constexpr (ks + 1 < KQ) __builtin_amdgcn_sched_barrier(0);
                  }
                  acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks % (PD + 1)][i], bw[ks % (PD + 1)][jj], acc[i][jj], 0, 0, 0);
              });
              S2VT_STAMP_AT(5);
              stage_c = stage_c + 1 == DM ? 0 : stage_c + 1;
              __syncthreads();
              S2VT_STAMP_AT(6);
          }
        }
      }
    } else
    if constexpr (VEC) {
      if (nchunks > 0) {
        // ---- interleaved loop.  One wave per SIMD is the normal occupancy of the step kernels, so whatever is issued
        // between two barriers but not BETWEEN two MFMAs runs with the matrix pipe idle (s_memtime stamps of a
        // sequential issue / compute / land form: load issue 600 + LDS stores 290 of 2600 clocks per chunk, gw80
        // tile).  Here the side work of a chunk is cut into A4+B4 pieces spliced after fixed MFMAs of the chunk's
        // list (each 16x16x4 fp32 MFMA leaves ~24 issue clocks free): first half = the raw-buffer loads of chunk
        // c+PF, middle = the counted vmcnt wait, second half = the LDS stores of chunk c+1 into the stage nobody
        // reads; fragments are read ahead in source order, fenced by sched_barrier (hipcc otherwise sinks the reads
        // to their first use) -- A four k-steps ahead (b128), B one.
        constexpr int NG4 = KQ / 4, MPK = TM * TN, NM = KQ * MPK, HALF = NM / 2;
        const i32x4 rsW = make_rsrc(g.W);
        const i32x4 rsA0 = make_rsrc(sp0), rsA1 = make_rsrc(sp1), rsA2 = make_rsrc(sp2);
        const int akq = (ltid % KQ) * 4;
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
            const int idx = ltid + i * NTL;
            if constexpr (BT) {
                const int cc = n0 + idx / KQ;                            // output column = row of W^T
                const bool ok = (B4 * NTL == BITEMS || idx < BITEMS) && cc < g.N;
                bvo[i] = ok ? (uint32_t)(cc * g.ldw + akq) * 4u : kOob;
                bkr[i] = akq;                                            // first k of this lane's float4 within a chunk
            } else {
                const int kr = idx / (BN / 4);
                const int col = (idx % (BN / 4)) * 4;
                const int grp = col / CG, cc = n0 + col % CG;
                const bool ok = (B4 * NTL == BITEMS || idx < BITEMS) && cc < g.N;
                bvo[i] = ok ? (uint32_t)(kr * g.ldw + grp * g.gstride + cc) * 4u : kOob;
                bkr[i] = kr;
            }
        }
        // The walk of the NEXT chunk to issue, as loop-carried scalars: segment, rows of k left in it, byte offsets.
        // They advance by BK per chunk; only a segment change (rare, wave-uniform branch) re-selects the descriptor
        // and the per-lane row offsets.  Past the end of the walk krem_ = 0: every lane is out of range and the ring
        // loads return zeros without touching memory.
        int wseg_ = -1, krem_ = 0;
        i32x4 rsA_ = rsA0;
        uint32_t soffA_ = 0u, soffW_ = 0u;
        uint32_t avo_[A4];
#pragma unroll
        for (int i = 0; i < A4; ++i) avo_[i] = kOob;
#define S2VT_WALK_ENTER()                                                                                \
        do {                                                                                             \
            ++wseg_;                                                                                     \
            while (wseg_ < 3 && (wseg_ == 0 ? nch0 : (wseg_ == 1 ? nch1 : nch2)) == 0) ++wseg_;          \
            if (wseg_ < 3) {                                                                             \
                rsA_ = wseg_ == 0 ? rsA0 : (wseg_ == 1 ? rsA1 : rsA2);                                   \
                krem_ = wseg_ == 0 ? slen0 : (wseg_ == 1 ? slen1 : slen2);                               \
                const int kw_ = (wseg_ == 0 ? skw0 : (wseg_ == 1 ? skw1 : skw2)) + kbeg;                 \
                soffA_ = 0u;                                                                             \
                soffW_ = (uint32_t)kw_ * (BT ? 1u : (uint32_t)g.ldw) * 4u;                               \
                _Pragma("unroll") for (int i_ = 0; i_ < A4; ++i_)                                        \
                    avo_[i_] = wseg_ == 0 ? avo0[i_] : (wseg_ == 1 ? avo1[i_] : avo2[i_]);               \
            } else {                                                                                     \
                krem_ = 0;                                                                               \
            }                                                                                            \
        } while (0)
#define S2VT_WALK_NEXT() /* after the pieces of one chunk have been issued */                            \
        do {                                                                                             \
            krem_ -= BK;                                                                                 \
            soffA_ += (uint32_t)BK * 4u;                                                                 \
            soffW_ += (uint32_t)BK * (BT ? 1u : (uint32_t)g.ldw) * 4u;                                   \
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
                const int idx = ltid + P * NTL;
                if (A4 * NTL == BM * KQ || idx < BM * KQ) a_store(buf, idx, qa[P]);
            } else {
                constexpr int i = P - A4;
                pin(qb[i]);
                const int idx = ltid + i * NTL;
                if (B4 * NTL == BITEMS || idx < BITEMS) b_store(buf, idx, qb[i]);
            }
        };
        auto splice = [](int p) constexpr { return ((2 * p + 1) * HALF) / (2 * LPC); };   // MFMA after which piece p of a half goes

        if constexpr (PW > 0) {
          // ---- loader / MFMA wave specialisation.  With one MFMA wave per SIMD every buffer load and LDS store the wave
          // issues itself stalls its MFMA stream for the length of the issue (stamps: ~85 clocks per load, a chunk
          // half of 20 MFMAs takes 870 instead of 450); here the PW loader waves carry the ring and the LDS stores,
          // and the MFMA waves issue nothing but fragment reads and MFMAs.  Same protocol as the interleaved loop:
          // one barrier per chunk, chunk c+1 is stored to the stage nobody reads while chunk c is multiplied.
          if (!mma_wave) {
            __builtin_amdgcn_s_setprio(3);     // few, short instructions that everything else waits for: ahead of the MFMA stream
            S2VT_WALK_ENTER();
            static_for<0, LPC>([&](auto p_) { constexpr int p = decltype(p_)::value; S2VT_PIECE_ISSUE(p, 0); });
            S2VT_WALK_NEXT();
            wait_vmcnt<0>();
            static_for<0, LPC>([&](auto p_) { land_piece(0, p_, ra[0], rb[0]); });
#pragma unroll
            for (int j = 1; j < PF; ++j) {
                static_for<0, LPC>([&](auto p_) { constexpr int p = decltype(p_)::value; S2VT_PIECE_ISSUE(p, j); });
                S2VT_WALK_NEXT();
            }
            __syncthreads();
            // Steady state, LDS stores BEFORE the loads: a wave's DS and VMEM instructions leave through one in-order
            // path, and an LDS store issued right behind five buffer loads sat there ~1000 clocks (stamps) -- behind the
            // previous iteration's loads, long gone by now, it does not wait.
            constexpr int WAITL = (PF - 2) * LPC;                  // chunks c+2 .. c+PF-1 may stay in flight
            int c = 0;
            bool more = true;
            while (more) {
#pragma unroll
                for (int j = 0; j < PF; ++j) {
                    if (more) {
                        wait_vmcnt<WAITL>();
                        S2VT_STAMP_AT(3);                          // (dev build) loader: ring wait
                        static_for<0, LPC>([&](auto p_) { land_piece((c + 1) & 1, p_, ra[(j + 1) % PF], rb[(j + 1) % PF]); });
                        S2VT_STAMP_AT(4);                          // loader: LDS stores
                        static_for<0, LPC>([&](auto p_) { constexpr int p = decltype(p_)::value; S2VT_PIECE_ISSUE(p, j); });
                        S2VT_WALK_NEXT();
                        S2VT_STAMP_AT(2);                          // loader: issue
                        __syncthreads();
                        S2VT_STAMP_AT(7);                          // loader: barrier
                        ++c;
                        more = c < nchunks;
                    }
                }
            }
            wait_vmcnt<0>();
#pragma unroll
            for (int j = 0; j < PF; ++j) {
#pragma unroll
                for (int i = 0; i < A4; ++i) pin(ra[j][i]);
#pragma unroll
                for (int i = 0; i < B4; ++i) pin(rb[j][i]);
            }
          } else {
            __syncthreads();
            S2VT_STAMP_AT(0);
            for (int c = 0; c < nchunks; ++c) {
                const int buf = c & 1;
                const float* a = As + buf * ABUF + a_frag;
                const float* b = Bs + buf * BBUF + b_frag;
                constexpr int PD = Cfg::PD;
                f32x4 a4[2][TM];
                float bv[PD + 1][TN];
                auto read_a = [&](auto g4_, f32x4 (&q)[TM]) __attribute__((always_inline)) {
                    constexpr int g4 = decltype(g4_)::value;
#pragma unroll
                    for (int i = 0; i < TM; ++i) q[i] = *reinterpret_cast<const f32x4*>(a + i * 16 * RS + ((g4 ^ a_swz) << 2));
                };
                auto read_b = [&](auto ks_, float (&q)[TN]) __attribute__((always_inline)) {
                    constexpr int ks = decltype(ks_)::value;
#pragma unroll
                    for (int jj = 0; jj < TN; ++jj) q[jj] = b[ks * 4 * SB + b_col(jj)];
                };
                read_a(std::integral_constant<int, 0>{}, a4[0]);
                static_for<0, (PD < KQ ? PD : KQ)>([&](auto k_) { read_b(k_, bv[decltype(k_)::value % (PD + 1)]); });
                S2VT_STAMP_AT(1);
                static_for<0, NM>([&](auto n_) {
                    constexpr int n = decltype(n_)::value, ks = n / MPK, r = n % MPK, i = r / TN, jj = r % TN;
                    if constexpr (r == 0) {
                        if constexpr (ks % 4 == 0 && ks / 4 + 1 < NG4)
                            read_a(std::integral_constant<int, ks / 4 + 1>{}, a4[(ks / 4 + 1) & 1]);
                        if constexpr (ks + PD < KQ) read_b(std::integral_constant<int, ks + PD>{}, bv[(ks + PD) % (PD + 1)]);
                        if constexpr (ks + 1 < KQ) __builtin_amdgcn_sched_barrier(0);
                    }
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[(ks / 4) & 1][i][ks % 4], bv[ks % (PD + 1)][jj], acc[i][jj], 0, 0, 0);
                });
                S2VT_STAMP_AT(5);
                __syncthreads();
                S2VT_STAMP_AT(6);
            }
          }
        } else {
        // prologue: chunk 0 -> stage 0; chunks 1 .. PF-1 in flight in ring slots 1 .. PF-1
        S2VT_WALK_ENTER();
        static_for<0, LPC>([&](auto p_) { constexpr int p = decltype(p_)::value; S2VT_PIECE_ISSUE(p, 0); });
        S2VT_WALK_NEXT();
        wait_vmcnt<0>();
        pin_epilogue_operands();
        static_for<0, LPC>([&](auto p_) { land_piece(0, p_, ra[0], rb[0]); });
#pragma unroll
        for (int j = 1; j < PF; ++j) {
            static_for<0, LPC>([&](auto p_) { constexpr int p = decltype(p_)::value; S2VT_PIECE_ISSUE(p, j); });
            S2VT_WALK_NEXT();
        }
        __syncthreads();
        S2VT_STAMP_AT(0);
        // Steady state: iteration c has stage c&1 = chunk c and ring slot (c+i)%PF = chunk c+i in flight (i = 1..PF-1);
        // it issues chunk c+PF into the slot chunk c came from and lands chunk c+1 in the other stage.  Unrolled by
        // PF so ring slots are compile-time constants; the only branch is the wave-uniform loop exit.
        int c = 0;
        bool more = true;
        while (more) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                if (more) {
                    const int buf = c & 1;
                    const float* a = As + buf * ABUF + a_frag;
                    const float* b = Bs + buf * BBUF + b_frag;
                    constexpr int PD = Cfg::PD;
                    f32x4 a4[2][TM];
                    float bv[PD + 1][TN];
                    f32x4 b4[2][TN];                                   // BT: B fragments in groups of four k-steps, like A
                    auto read_a = [&](auto g4_, f32x4 (&q)[TM]) __attribute__((always_inline)) {
                        constexpr int g4 = decltype(g4_)::value;
#pragma unroll
                        for (int i = 0; i < TM; ++i) q[i] = *reinterpret_cast<const f32x4*>(a + i * 16 * RS + ((g4 ^ a_swz) << 2));
                    };
                    auto read_b = [&](auto ks_, float (&q)[TN]) __attribute__((always_inline)) {
                        constexpr int ks = decltype(ks_)::value;
#pragma unroll
                        for (int jj = 0; jj < TN; ++jj) q[jj] = b[ks * 4 * SB + b_col(jj)];
                    };
                    auto read_bg = [&](auto g4_, f32x4 (&q)[TN]) __attribute__((always_inline)) {
                        constexpr int g4 = decltype(g4_)::value;
#pragma unroll
                        for (int jj = 0; jj < TN; ++jj) q[jj] = *reinterpret_cast<const f32x4*>(b + b_col(jj) * KQ + ((g4 ^ a_swz) << 2));
                    };
                    read_a(std::integral_constant<int, 0>{}, a4[0]);
                    if constexpr (BT) read_bg(std::integral_constant<int, 0>{}, b4[0]);
                    else static_for<0, (PD < KQ ? PD : KQ)>([&](auto k_) { read_b(k_, bv[decltype(k_)::value % (PD + 1)]); });
                    S2VT_STAMP_AT(1);                              // (dev build) top-of-chunk fragment latency
                    static_for<0, NM>([&](auto n_) {
                        constexpr int n = decltype(n_)::value, ks = n / MPK, r = n % MPK, i = r / TN, jj = r % TN;
                        if constexpr (r == 0) {
                            if constexpr (ks % 4 == 0 && ks / 4 + 1 < NG4)
                                read_a(std::integral_constant<int, ks / 4 + 1>{}, a4[(ks / 4 + 1) & 1]);
                            if constexpr (BT) {
                                if constexpr (ks % 4 == 0 && ks / 4 + 1 < NG4)
                                    read_bg(std::integral_constant<int, ks / 4 + 1>{}, b4[(ks / 4 + 1) & 1]);
                            } else if constexpr (ks + PD < KQ) {
                                read_b(std::integral_constant<int, ks + PD>{}, bv[(ks + PD) % (PD + 1)]);
                            }
                            if constexpr (ks + 1 < KQ) __builtin_amdgcn_sched_barrier(0);
                        }
                        if constexpr (n == HALF) {
                            S2VT_STAMP_AT(2);                      // first half: MFMAs + load issue
                            wait_vmcnt<WAITN>();
                            S2VT_STAMP_AT(3);                      // ring wait
                        }
                        if constexpr (BT)
                            acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[(ks / 4) & 1][i][ks % 4], b4[(ks / 4) & 1][jj][ks % 4], acc[i][jj], 0, 0, 0);
                        else
                            acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[(ks / 4) & 1][i][ks % 4], bv[ks % (PD + 1)][jj], acc[i][jj], 0, 0, 0);
                        static_for<0, LPC>([&](auto p_) {
                            constexpr int p = decltype(p_)::value;
                            if constexpr (splice(p) == n) S2VT_PIECE_ISSUE(p, j);
                            if constexpr (HALF + splice(p) == n) land_piece((c + 1) & 1, p_, ra[(j + 1) % PF], rb[(j + 1) % PF]);
                        });
                    });
                    S2VT_STAMP_AT(5);                              // second half: MFMAs + LDS stores
                    S2VT_WALK_NEXT();
                    __syncthreads();
                    S2VT_STAMP_AT(6);                              // barrier
                    ++c;
                    more = c < nchunks;
                }
            }
        }
        // The ring's last PF-1 chunks (beyond the walk, never landed) are still in flight when the loop exits.  Wait for
        // them AND keep every ring register formally alive until after that wait: an asm load's destination is
        // "written" at the asm statement as far as hipcc knows, so a slot that is never read again is free at once,
        // and hipcc hands it to epilogue values (pure register code may be scheduled above an asm volatile) that the
        // late-arriving load data then overwrites -- seen as wrong dwords / wild addresses with cold caches.
        wait_vmcnt<0>();
#pragma unroll
        for (int j = 0; j < PF; ++j) {
#pragma unroll
            for (int i = 0; i < A4; ++i) pin(ra[j][i]);
#pragma unroll
            for (int i = 0; i < B4; ++i) pin(rb[j][i]);
        }
        S2VT_STAMP_AT(7);
        }
#undef S2VT_WALK_ENTER
#undef S2VT_WALK_NEXT
#undef S2VT_PIECE_ISSUE
      }
    } else {
        // ---- odd shapes (unaligned pointers / sizes not multiples of 4: tests and tiny problems): the same walk with
        // ordinary compiler-scheduled scalar loads, zero-filled in registers, one chunk at a time.
        auto load_chunk = [&](int koff, const float* abase, int sk, int kw, const int (&aro)[A4], f32x4 (&qa)[A4], f32x4 (&qb)[B4]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < A4; ++i) {
                const int idx = ltid + i * NTL;
                const int k = koff + (idx % KQ) * 4;
                const int ro = aro[i];
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool ok = ro >= 0 && k + e < sk;
                    const float x = abase[ok ? ro + k + e : 0];
                    v[e] = ok ? x : 0.f;
                }
                qa[i] = v;
            }
#pragma unroll
            for (int i = 0; i < B4; ++i) {
                const int idx = ltid + i * NTL;
                f32x4 v;
                if constexpr (BT) {
                    const int cc = n0 + idx / KQ;
                    const int k = koff + (idx % KQ) * 4;
                    const bool rok = (B4 * NTL == BITEMS || idx < BITEMS) && cc < g.N;
                    const float* wrow = g.W + (size_t)(rok ? cc : 0) * g.ldw + kw;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool ok = rok && k + e < sk;
                        const float x = wrow[ok ? k + e : 0];
                        v[e] = ok ? x : 0.f;
                    }
                } else {
                    const int kr = idx / (BN / 4);
                    const int col = (idx % (BN / 4)) * 4;
                    const int grp = col / CG, cc = n0 + col % CG;
                    const int k = koff + kr;
                    const bool kok = (B4 * NTL == BITEMS || idx < BITEMS) && k < sk;
                    const float* wrow = g.W + (size_t)(kw + (kok ? k : 0)) * g.ldw + grp * g.gstride;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool ok = kok && cc + e < g.N;
                        const float x = wrow[ok ? cc + e : 0];
                        v[e] = ok ? x : 0.f;
                    }
                }
                qb[i] = v;
            }
        };
        const int cum0 = nch0, cum1 = nch0 + nch1;
        for (int c = 0; c < nchunks; ++c) {
            const int sidx_ = (c >= cum0 ? 1 : 0) + (c >= cum1 ? 1 : 0);
            const int cstart_ = sidx_ == 0 ? 0 : (sidx_ == 1 ? cum0 : cum1);
            const float* abase_ = sidx_ == 0 ? sp0 : (sidx_ == 1 ? sp1 : sp2);
            const int sk_ = sidx_ == 0 ? slen0 : (sidx_ == 1 ? slen1 : slen2);
            const int kw_ = (sidx_ == 0 ? skw0 : (sidx_ == 1 ? skw1 : skw2)) + kbeg;
            int aro_[A4];
#pragma unroll
            for (int i_ = 0; i_ < A4; ++i_) aro_[i_] = sidx_ == 0 ? aoff0[i_] : (sidx_ == 1 ? aoff1[i_] : aoff2[i_]);
            if (ltid >= 0) load_chunk((c - cstart_) * BK, abase_, sk_, kw_, aro_, ra[0], rb[0]);
            __syncthreads();                                       // everyone is done reading the previous chunk
            if (ltid >= 0) {
#pragma unroll
                for (int i = 0; i < A4; ++i) {
                    const int idx = ltid + i * NTL;
                    if (A4 * NTL == BM * KQ || idx < BM * KQ) a_store(0, idx, ra[0][i]);
                }
#pragma unroll
                for (int i = 0; i < B4; ++i) {
                    const int idx = ltid + i * NTL;
                    if (B4 * NTL == BITEMS || idx < BITEMS) b_store(0, idx, rb[0][i]);
                }
            }
            __syncthreads();
            const float* a = As + a_frag;
            const float* b = Bs + b_frag;
            if (mma_wave)
#pragma unroll
            for (int ks = 0; ks < KQ; ++ks) {
                float av[TM], bw[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) av[i] = a[i * 16 * RS + ((((ks >> 2) ^ a_swz) << 2) | (ks & 3))];
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bw[j] = BT ? b[b_col(j) * KQ + ((((ks >> 2) ^ a_swz) << 2) | (ks & 3))] : b[ks * 4 * SB + b_col(j)];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bw[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    // ------------------------------------------------------------------ epilogues
    if constexpr (PW > 0 && EPI != EPI_LSTM_GW) {
        if (!mma_wave) return;                 // loader waves hold no results (an ended wave no longer counts at barriers)
    }
    if constexpr (EPI == EPI_STORE) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cc = n0 + (wn * TNG + j % TNG) * 16 + l15;
            const int col = (j / TNG) * g.gstride + cc;
            if (cc >= g.N) continue;
            const float bj = ep_bias[j];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + (wm * TM + i) * 16 + lq * 4 + r;
                    if (m < MM) {
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
                    if (m >= MM) continue;
                    const int mo = orow(m);
                    const float zi = acc[i][0 * TNG + jj][r] + bi;
                    const float zj = acc[i][1 * TNG + jj][r] + bj;
                    const float zf = acc[i][2 * TNG + jj][r] + bf;
                    const float zo = acc[i][3 * TNG + jj][r] + bo;
                    const float si = dm_sigmoidf(zi);
                    const float tj = dm_tanhf(zj);
                    const float sf = dm_sigmoidf(zf + 1.0f);
                    const float so = dm_sigmoidf(zo);
                    const size_t o = (size_t)mo * H + u;
                    const size_t op = (size_t)(g.cprev_rowmod > 0 ? mo % g.cprev_rowmod : mo) * H + u;
                    const float t1 = g.c_prev[op] * sf;
                    const float t2 = si * tj;
                    const float c = t1 + t2;
                    const float h = dm_tanhf(c) * so;
                    g.c_new[o] = c;
                    g.h_new[o] = h;
                    if (g.out) {
                        float ov = h;
                        if (g.keep < 1.0f) {
                            const float k01 = dropout_keep01(g.seed_lo, g.seed_hi, (uint32_t)g.video_id[mo],
                                                             (uint32_t)g.sample_id[mo], g.drop_code, (uint32_t)u, g.keep);
                            ov = (h / g.keep) * k01;
                        }
                        g.out[o] = ov;
                    }
                    if (g.gates) {
                        float* gp = g.gates + (size_t)mo * 4 * H + u;
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
        if (mma_wave)
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
            if (m >= MM || u >= H) continue;
            const int mo = orow(m);
            const float* z = smem + row * ZS + uu;
            const float zi = z[0] + g.bias[u];
            const float zj = z[CG] + g.bias[H + u];
            const float zf = z[2 * CG] + g.bias[2 * H + u];
            const float zo = z[3 * CG] + g.bias[3 * H + u];
            const float si = dm_sigmoidf(zi);
            const float tj = dm_tanhf(zj);
            const float sf = dm_sigmoidf(zf + 1.0f);
            const float so = dm_sigmoidf(zo);
            const size_t o = (size_t)mo * H + u;
            const size_t op = (size_t)(g.cprev_rowmod > 0 ? mo % g.cprev_rowmod : mo) * H + u;
            const float t1 = g.c_prev[op] * sf;
            const float t2 = si * tj;
            const float c = t1 + t2;
            const float h = dm_tanhf(c) * so;
            g.c_new[o] = c;
            g.h_new[o] = h;
            if (g.out) {
                float ov = h;
                if (g.keep < 1.0f) {
                    const float k01 = dropout_keep01(g.seed_lo, g.seed_hi, (uint32_t)g.video_id[mo], (uint32_t)g.sample_id[mo],
                                                     g.drop_code, (uint32_t)u, g.keep);
                    ov = (h / g.keep) * k01;
                }
                g.out[o] = ov;
            }
            if (g.gates) {
                float* gp = g.gates + (size_t)mo * 4 * H + u;
                gp[0] = si; gp[H] = tj; gp[2 * H] = sf; gp[3 * H] = so;
            }
        }
    } else {  // EPI_PICK
        // Column tiles start at multiples of 16, so the quad l15 = 4q .. 4q+3 holds columns 4Q .. 4Q+3 of the
        // SAME four rows (lq*4 + r): quad lane e draws the Philox block of row r = e, the words are exchanged
        // with DPP (detmath.h).  All 64 lanes take part in every exchange (no divergence around it).
        const uint32_t e4 = (uint32_t)l15 & 3u;
        S2VT_STAMP_AT(9);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mrow = m0 + (wm * TM + i) * 16 + lq * 4;
            int sid[4], vid[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sid[r] = ep_sid[i][r];
                vid[r] = ep_vid[i][r];
            }
            const int sid_own = e4 == 0 ? sid[0] : e4 == 1 ? sid[1] : e4 == 2 ? sid[2] : sid[3];
            const int vid_own = e4 == 0 ? vid[0] : e4 == 1 ? vid[1] : e4 == 2 ? vid[2] : vid[3];
#ifdef S2VT_PICK_EXACT_ALL     // dev A/B: the exact noise for every element (the form before the two-tier screen)
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
                    if (m < MM && col < g.N) {
                        float v = acc[i][j][r] + ep_bias[j];
                        if (g.logits_out) g.logits_out[(size_t)orow(m) * g.ldc + col] = v;
                        if (sid[r] >= 0) v = v + gumbel_from_word(word[r]);
                        v = v + 0.0f;  // -0 -> +0 so that the integer order equals the float order
                        if (!have[r] || v > best[r]) { best[r] = v; bidx[r] = (uint32_t)col; have[r] = true; }
                    }
                }
            }
#else
            float best[4];
            uint32_t bidx[4];
            bool have[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { best[r] = 0.0f; bidx[r] = 0xFFFFFFFFu; have[r] = false; }
            // Two-tier Gumbel-max (exact result, ~1/3 of the noise arithmetic).  The exact key of an element is
            //     v = (acc + bias) + gumbel_from_word(word);  v = v + 0.0f
            // -- two Cephes logs per element, 8 us of VALU per launch when done for all 24 elements of a lane.  Tier 1 keys
            // every element with a FAST Gumbel value (two v_log_f32, |fast - exact| << kGumbelScreenMargin) and takes the
            // row maximum over the workgroup tile's columns this wave holds.  An element whose fast key is more than
            // 2 x margin below that maximum cannot be the exact maximum of the row (the fast key of the exact winner is
            // within the margin of its exact key, and the fast maximum is at least that).  Tier 2 evaluates the EXACT
            // expression -- the same instruction sequence as before -- only for the survivors: per row typically ONE
            // element of the lane's TN (its fast-best), any other survivor in a guarded rare path.  Ties and order:
            // (v > best) or (v == best and col < bidx), i.e. the lowest column among equal keys, as the ascending scan gave.
            float v0[TN][4], ka[TN][4];
            uint32_t wd[TN][4];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + (wn * TN + j) * 16 + l15;
                u32x4 blk = {0u, 0u, 0u, 0u};
#ifndef S2VT_PICK_NONOISE
                if (sid_own >= 0)
                    blk = philox4x32_10((uint32_t)col >> 2, (uint32_t)vid_own, (uint32_t)sid_own, (uint32_t)g.step, g.seed_lo,
                                        g.seed_hi);
#endif
                wd[j][0] = quad_word_from<0>(blk, e4);
                wd[j][1] = quad_word_from<1>(blk, e4);
                wd[j][2] = quad_word_from<2>(blk, e4);
                wd[j][3] = quad_word_from<3>(blk, e4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mrow + r;
                    const bool ok = m < MM && col < g.N;
                    const float v = acc[i][j][r] + ep_bias[j];
                    v0[j][r] = v;
                    if (ok && g.logits_out) g.logits_out[(size_t)orow(m) * g.ldc + col] = v;
                    float k = v;
#ifndef S2VT_PICK_NONOISE       // (dev ablation)
                    if (sid[r] >= 0) k = v + gumbel_fast_from_word(wd[j][r]);
#endif
                    ka[j][r] = ok ? k : -__builtin_inff();
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // fast row maximum: this lane's TN columns, then the 16 lanes that hold the row
                float mx = ka[0][r];
                int js = 0;
#pragma unroll
                for (int j = 1; j < TN; ++j)
                    if (ka[j][r] > mx) { mx = ka[j][r]; js = j; }
                float rmx = mx;
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) rmx = fmaxf(rmx, __shfl_xor(rmx, off, 64));
                const float thr = rmx - 2.0f * kGumbelScreenMargin;
                // the lane's fast-best element of this row
                float vs = v0[0][r];
                uint32_t ws = wd[0][r];
#pragma unroll
                for (int j = 1; j < TN; ++j)
                    if (js == j) { vs = v0[j][r]; ws = wd[j][r]; }
                const int cols = n0 + (wn * TN + js) * 16 + l15;
                const bool okr = mrow + r < MM;
                if (okr && cols < g.N && !(mx < thr)) {
                    float v = vs;
#ifndef S2VT_PICK_NONOISE
                    if (sid[r] >= 0) v = v + gumbel_from_word(ws);
#endif
                    v = v + 0.0f;  // -0 -> +0 so that the integer order equals the float order
                    best[r] = v; bidx[r] = (uint32_t)cols; have[r] = true;
                }
                // rare: a second element of this lane's row within the margin
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = n0 + (wn * TN + j) * 16 + l15;
                    const bool extra = okr && col < g.N && j != js && !(ka[j][r] < thr);
                    if (__any(extra)) {
                        if (extra) {
                            float v = v0[j][r];
                            if (sid[r] >= 0) v = v + gumbel_from_word(wd[j][r]);
                            v = v + 0.0f;
                            if (!have[r] || v > best[r] || (v == best[r] && (uint32_t)col < bidx[r])) { best[r] = v; bidx[r] = (uint32_t)col; have[r] = true; }
                        }
                    }
                }
            }
#endif
            S2VT_STAMP_AT(10);                     // (dev build) noise + per-lane argmax
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
#ifndef S2VT_PICK_NOATOMIC      // (dev ablation)
                if (l15 == 0 && m < MM && key != 0ull) atomicMax(&g.pick[(size_t)orow(m) * (g.pick_stride > 0 ? g.pick_stride : 1)], key);
#else
                if (l15 == 0 && m < MM && key == 1ull) g.pick[(size_t)orow(m) * (g.pick_stride > 0 ? g.pick_stride : 1)] = key;
#endif
            }
            S2VT_STAMP_AT(11);                     // (dev build) lane reduction + atomics
        }
    }
#ifdef S2VT_STAMP
    S2VT_STAMP_AT(8);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 14; ++i) atomicAdd(&s2vt_stamp_acc[i], st_acc[i]);
        atomicAdd(&s2vt_stamp_acc[14], 1ull);
        atomicAdd(&s2vt_stamp_acc[15], (unsigned long long)nchunks);
    }
#endif
}

}  // namespace s2vt
