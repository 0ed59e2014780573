// chain.hip -- a whole BasicLSTMCell recurrence (T steps, M <= 384 rows) in ONE launch: the unroll of tf_s2vt.py:113-153
// for a cell whose only step-dependent input is a partial that is known before the loop (LSTM1 over a video: frame
// rows hoisted, then the zero padding; LSTM2 in build_model once [out1 ; embed(word)] @ W2 has been hoisted).
//
// Why its own kernel.  At M <= 64 a step is 0.5 GFLOP: the per-step launch (gemm_mfma.h, gw16 tile) spends ~15 us on it
// whatever M is, because every launch re-streams the 16 MB recurrent weight block from L2, refills its pipeline and
// drains it.  Here the recurrent rows of W never move after the prologue:
//   * workgroup j owns 4 hidden units = 16 gate columns; its [H x 16] slice of W (64 KB fp32 at H = 1000) is gathered
//     into LDS once, in MFMA B-fragment order (one ds_read_b128 = the operands of four k-steps);
//   * the state h_t goes from every workgroup to every workgroup through L2 in MFMA A-fragment order
//     ([row tile][k group][lane][4]: 1 KB per wave-instruction, fully coalesced), so A fragments are loaded straight into
//     registers -- a ring of 32 groups (32 KB per wave) in flight -- and never touch LDS;
//   * c_t stays in a register of the lane that owns (row, unit) for all T steps; the gates of a unit meet through a
//     256-float per-wave LDS tile;
//   * per step one grid-wide hand-off: every workgroup stores its 1 KB slice of h_t write-through (sc1), drains, adds to
//     its XCD shard of a counter; one wave per workgroup polls the 8 shards (sc1 loads) and every load of h_t is an sc1
//     load -- the measured sc1 form of MI355X_MICROARCH.md "Valid forms", row 1 (one lane of each storing workgroup
//     signals for all its stores; poll of every shard; other waves load behind the workgroup barrier).
// Arithmetic contract unchanged: each pre-activation is cinit (+) the ascending-k fp32 chain over the
// recurrent rows, v_mfma_f32_16x16x4_f32 in k order; the pointwise part is the EPI_LSTM expression sequence, so states,
// gates and dropped outputs are bit-identical to T per-step launches (tests/test_gpu_chain.py).
// All workgroups must be co-resident (grid = H / 4 <= CU count, one 256-thread workgroup per CU).  What guards that:
//   * the launcher checks hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs >= grid per configuration and device;
//   * persistent launches of one process never overlap: a launch on another stream first waits for the event recorded
//     behind the previous one (two half-resident grids would starve each other);
//   * every spin is bounded; a timeout raises a host-mapped counter AND a device-resident fault word.  The fault is
//     sticky until acknowledged (s2vt_chain_ack): every later library call that would launch a recurrence or update the
//     variables returns S2VT_E_CHAIN_TIMEOUT, and adam_tf_kernel launches already queued behind the fault SKIP their
//     update on the device -- a starved recurrence can produce garbage activations, never garbage variables.
// Another PROCESS on the same GPU running its own persistent grid cannot be seen from here: the GPU must be exclusive to
// the process, or S2VT_CHAIN=0.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "chain_common.h"

namespace s2vt {

namespace {
// NC = 16-column tiles of gate columns per workgroup (4 NC hidden units).  NC = 1: the rows are not split, a wave owns TMW
// row tiles and streams their whole state image every step -- at M >= 128 that stream (M*H*4 bytes per CU per step through
// the CU's L2 port, ~15 B/clk) takes longer than the MFMAs.  NC = 2: each workgroup owns 8 units and HALF the row tiles
// (g.tpp per part, grid = 2 * H/8), so a CU streams half the image for the same number of MFMAs.
template <int NG, int TMW, int NC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_chain_kernel(const ChainArgs g)
{
    constexpr int ZS = 20;                                     // z tile row stride (floats): 16-byte aligned rows, conflict-light
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                                          // [NG][NC][64 lanes][4]: B fragments of this workgroup's 16 NC gate columns
    const int tid = threadIdx.x, lane = tid & 63;
    const int pwave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* zb = smem + NG * NC * 256 + pwave * (16 * ZS);
    // which quarter of the rows this wave takes rotates with the workgroup: neighbouring CUs then walk DIFFERENT lines of
    // the shared state image at any moment instead of all hammering the same L2 channel (speed only)
    const int wave = (pwave + (int)blockIdx.x) & 3;
    const int l15 = lane & 15, lq = lane >> 4;
    const int H = g.H, M = g.M, T = g.T;
    const int cg = (int)blockIdx.x % g.ncg, rp = (int)blockIdx.x / g.ncg;     // column group, row part
    const int u0 = cg * 4 * NC;
    const int nwg = gridDim.x;
    const int tb = rp * g.tpp + wave * TMW;                    // first row tile of this wave
    bool tok[TMW];                                             // tile i is this wave's to compute (inside its part and inside M)
#pragma unroll
    for (int i = 0; i < TMW; ++i) tok[i] = wave * TMW + i < g.tpp && (tb + i) * 16 < M;

    // ---- this workgroup's slice of the recurrent rows -> LDS, once.  Column cc = uu * 4 + gate of column tile c is W column
    // gate * H + u0 + 4c + uu; element (k, cc) goes to group k / 16, tile c, lane (k % 4) * 16 + cc, component (k % 16) / 4.
    for (int idx = tid; idx < NG * 16 * 4 * NC; idx += 256) {
        const int c = idx % NC, kg = idx / NC;
        const int k = kg >> 2, gt = kg & 3;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < H) v = *reinterpret_cast<const f32x4*>(g.W + (size_t)(g.kw0 + k) * g.ldw + (size_t)gt * H + u0 + 4 * c);
        const int j = k >> 4, e = (k & 15) >> 2, kq = k & 3;
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) Wl[(((j * NC + c) * 64 + kq * 16 + uu * 4 + gt) << 2) + e] = v[uu];
    }

    // ---- the (row, unit) pairs this lane finishes at every step: one per (column tile, row tile) of its wave
    const int rt = lane >> 2, uu = lane & 3;                   // row within a 16-row tile, unit within the 4
    const int row0 = tb * 16 + rt;                             // row of tile 0; tile i is 16 i further
    float bi[NC], bj[NC], bf[NC], bo[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int u = u0 + 4 * c + uu;
        bi[c] = g.bias[u]; bj[c] = g.bias[H + u]; bf[c] = g.bias[2 * H + u]; bo[c] = g.bias[3 * H + u];
    }
    float c_reg[NC][TMW];
    uint32_t vid[TMW], sid[TMW];
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
        const int row = row0 + 16 * i;
        const bool rok = tok[i] && row < M;
#pragma unroll
        for (int c = 0; c < NC; ++c) c_reg[c][i] = (rok && g.c0) ? g.c0[(size_t)row * H + u0 + 4 * c + uu] : 0.0f;
        vid[i] = (g.keep < 1.0f && rok) ? (uint32_t)g.video_id[row] : 0u;
        sid[i] = (g.keep < 1.0f && rok) ? (uint32_t)g.sample_id[row] : 0u;
    }
    // where this lane's h value of (column tile c, row tile 0) sits in the A-fragment image (tile i: + i * NG * 256 floats):
    // k = u -> group u / 16, lane (u % 4) * 16 + rt, component (u % 16) / 4
    size_t a_own[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int u = u0 + 4 * c + uu;
        a_own[c] = ((size_t)(tb * NG + (u >> 4)) * 64 + (size_t)((u & 3) * 16 + rt)) * 4 + ((u & 15) >> 2);
    }
    float* const abuf0 = g.abuf;
    float* const abuf1 = g.abuf + (size_t)g.img_tiles * NG * 256;

    GridSync gs{(gu32*)g.sync, g.status, g.fault, g.spin_limit, nwg, false};
    auto arrive = [&]() __attribute__((always_inline)) { gs.arrive(tid); };
    auto wait_all = [&](unsigned arrival) __attribute__((always_inline)) { gs.wait_all(arrival, pwave, lane); };

    // ---- arrival 0: h_0 in fragment order (the image is zero-filled by the launcher: rows >= M and k >= H stay zero)
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
        const int row = row0 + 16 * i;
        if (tok[i] && row < M) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float h0 = g.h0 ? g.h0[(size_t)row * H + u0 + 4 * c + uu] : 0.0f;
                __hip_atomic_store((gu32*)(abuf0 + a_own[c] + (size_t)i * NG * 256), __float_as_uint(h0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // the carried partial of step 0 (accumulator layout: lane (column l15, row group lq) holds rows lq*4 + r of a tile)
    float ci[NC][TMW][4];
    auto load_cinit = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ccol = (l15 & 3) * H + u0 + 4 * c + (l15 >> 2);      // W / cinit column of column l15 of tile c
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = (tb + i) * 16 + lq * 4 + r;
                    ci[c][i][r] = (g.cinit && t < g.cinit_steps && tok[i] && m < M) ? g.cinit[(size_t)t * g.cinit_tstride + (size_t)m * g.ldcinit + ccol] : 0.0f;
                }
        }
    };
    load_cinit(0);
    arrive();
    // per-lane byte offset of the A-fragment loads of tile i; a tile that is not this wave's reads nothing (out of range: zeros)
    int voff[TMW];
#pragma unroll
    for (int i = 0; i < TMW; ++i) voff[i] = tok[i] ? lane * 16 : (int)0x80000000u;

    for (int t = 0; t < T; ++t) {
        f32x4 acc[NC][TMW];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                acc[c][i] = f32x4{ci[c][i][0], ci[c][i][1], ci[c][i][2], ci[c][i][3]};
                asm volatile("" : "+v"(acc[c][i]));            // the partial is in registers before the ring below is issued
            }
        wait_all((unsigned)t);
        // ---- A fragments straight into registers.  Ring of RING groups (x TMW row tiles): group j + RING is issued into
        // group j's registers as soon as its MFMAs have read them.
        const float* acur = (t & 1) ? abuf1 : abuf0;
        const __amdgpu_buffer_rsrc_t rsA =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(acur + (size_t)tb * NG * 256), 0, TMW * NG * 1024, 0x00020000);
        constexpr int RING0 = TMW == 1 ? 32 : 40 / TMW;        // <= 160 ring registers
        constexpr int RING = NG < RING0 ? NG : RING0;
        f32x4 a[RING][TMW];
        static_for<0, RING>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            static_for<0, TMW>([&](auto i_) { constexpr int i = decltype(i_)::value; a[j][i] = bload16_sc1(rsA, voff[i], (i * NG + j) * 1024); });
        });
        __builtin_amdgcn_sched_barrier(0);
        const f32x4* bl = reinterpret_cast<const f32x4*>(Wl) + lane;
        constexpr int PB = NG < 4 ? NG : 4;                    // B fragments read PB groups ahead
        f32x4 b[PB][NC];
        static_for<0, PB>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
#pragma unroll
            for (int c = 0; c < NC; ++c) b[j][c] = bl[(j * NC + c) * 64];
        });
        static_for<0, NG>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            f32x4 bj4[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) bj4[c] = b[j % PB][c];
            if constexpr (j + PB < NG) {
#pragma unroll
                for (int c = 0; c < NC; ++c) b[j % PB][c] = bl[((j + PB) * NC + c) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, 4>([&](auto e_) {
                constexpr int e = decltype(e_)::value;
                static_for<0, NC>([&](auto c_) {
                    constexpr int c = decltype(c_)::value;
                    static_for<0, TMW>([&](auto i_) {
                        constexpr int i = decltype(i_)::value;
                        acc[c][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j % RING][i][e], bj4[c][e], acc[c][i], 0, 0, 0);
                    });
                });
            });
            if constexpr (j + RING < NG) {
                __builtin_amdgcn_sched_barrier(0);             // the refill stays behind the MFMAs that read these registers, and in place
                static_for<0, TMW>([&](auto i_) { constexpr int i = decltype(i_)::value; a[j % RING][i] = bload16_sc1(rsA, voff[i], (i * NG + j + RING) * 1024); });
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        // ---- per (column tile, row tile): the gates of a unit meet through the wave's LDS tile z[row][uu*4 + gate];
        // BasicLSTMCell pointwise (gate order i, j, f, o; forget_bias 1.0 added at run time) -- the EPI_LSTM expressions
        float hv[NC][TMW], siv[NC][TMW], tjv[NC][TMW], sfv[NC][TMW], sov[NC][TMW];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) zb[(lq * 4 + r) * ZS + l15] = acc[c][i][r];
                __builtin_amdgcn_wave_barrier();
                const f32x4 z = *reinterpret_cast<const f32x4*>(zb + rt * ZS + uu * 4);
                __builtin_amdgcn_wave_barrier();
                const float zi = z[0] + bi[c], zj = z[1] + bj[c], zf = z[2] + bf[c], zo = z[3] + bo[c];
                const float si = dm_sigmoidf(zi);
                const float tj = dm_tanhf(zj);
                const float sf = dm_sigmoidf(zf + 1.0f);
                const float so = dm_sigmoidf(zo);
                const float t1 = c_reg[c][i] * sf;
                const float t2 = si * tj;
                const float cc = t1 + t2;
                hv[c][i] = dm_tanhf(cc) * so;
                c_reg[c][i] = cc;
                siv[c][i] = si; tjv[c][i] = tj; sfv[c][i] = sf; sov[c][i] = so;
            }
        // The hand-off first: h_t write-through, drained and signalled BEFORE the history stores, which nobody in this
        // launch reads -- they complete under the other workgroups' arrival (and only have to by the end of the kernel).
        if (t + 1 < T) {
#pragma unroll
            for (int i = 0; i < TMW; ++i)
                if (tok[i] && row0 + 16 * i < M) {
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        __hip_atomic_store((gu32*)(((t & 1) ? abuf0 : abuf1) + a_own[c] + (size_t)i * NG * 256), __float_as_uint(hv[c][i]),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            arrive();
        }
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
            const int row = row0 + 16 * i;
            if (!tok[i] || row >= M) continue;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int u = u0 + 4 * c + uu;
                const size_t o = (size_t)row * H + u;
                g.C[(size_t)(t + 1) * g.state_tstride + o] = c_reg[c][i];
                g.Hh[(size_t)(t + 1) * g.state_tstride + o] = hv[c][i];
                if (g.out) {
                    float ov = hv[c][i];
                    if (g.keep < 1.0f)
                        ov = (hv[c][i] / g.keep) * dropout_keep01(g.seed_lo, g.seed_hi, vid[i], sid[i], g.drop_code0 + (uint32_t)t, (uint32_t)u, g.keep);
                    g.out[(size_t)t * g.out_tstride + o] = ov;
                }
                if (g.gates) {
                    float* gp = g.gates + (size_t)t * g.gates_tstride + (size_t)row * 4 * H + u;
                    gp[0] = siv[c][i]; gp[H] = tjv[c][i]; gp[2 * H] = sfv[c][i]; gp[3 * H] = sov[c][i];
                }
            }
        }
        if (t + 1 < T) load_cinit(t + 1);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// The forward recurrence above 256 rows with the recurrent weights in REGISTERS ("chain4", round 3).  At M = 320 the
// two-part form above spends 34.7 us per step against 16.7 us of MFMAs: each CU streams half the state image (640 KB) per step
// through its L2 port, 10 row tiles split 3/3/2/2 over the waves, and a wave finishes up to six (row tile, column tile)
// pointwise blocks in a row.  The LDS cannot hold more than 8 units' worth of W -- but the register file can: at one wave per
// SIMD a wave has 512 VGPRs, and the B fragments of ONE 16-column tile over all of K = 4 NG steps are 4 NG <= 256 of them.
//   * workgroup (j, part): 16 hidden units = 4 column tiles, wave w keeps column tile w (units 4j+... of all four gates) in
//     registers for the whole launch; the rows are cut into FOUR parts (grid = 4 ceil(H/16) = 252 at H = 1000), so a CU
//     streams a quarter of the state image per step (320 KB at M = 320);
//   * the four waves need the SAME A fragments, so the part's slice of the image goes global -> LDS once per CU
//     (`buffer_load_dwordx4 ... lds sc1`, 1 KB per instruction straight into fragment order, a ring of chunks of CG k-groups;
//     every wave issues a quarter of each chunk) and each wave reads its fragments with one ds_read_b128 per row tile and
//     k-group; one workgroup barrier per chunk;
//   * every wave runs TPP row tiles x 1 column tile: the MFMAs are balanced (5 x 256 per step at M = 320 = 17 us), the
//     pointwise part is TPP blocks per wave.
// Same arithmetic: each pre-activation is cinit (+) the ascending-k chain over the recurrent rows (k-groups in order, the
// four k-steps of a group in order), the pointwise part is the expression sequence of the kernel above -- bit-identical
// states, gates and outputs (tests/test_gpu_chain.py).
#ifdef S2VT_C4_STAMP
__device__ unsigned long long c4_stamp_acc[12 * 8];
#endif
template <int NG, int TPP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_chain4_kernel(const ChainArgs g)
{
    constexpr int ZS = 20;
#ifndef S2VT_C4_CG
#define S2VT_C4_CG 8
#endif
#ifndef S2VT_C4_NBUF
#define S2VT_C4_NBUF 3
#endif
    constexpr int CG = S2VT_C4_CG;                             // k-groups per chunk
    constexpr int NCH = NG / CG;                               // chunks per step
    constexpr int NBUF = S2VT_C4_NBUF;                         // LDS chunk buffers (NBUF - 1 chunks in flight)
    constexpr int CHF = TPP * CG * 256;                        // floats per chunk buffer
    constexpr int DPW = (TPP * CG + 3) / 4;                    // DMA instructions per wave per chunk
    static_assert(NG % CG == 0 && NCH >= NBUF, "chunking");
    static_assert((NCH - 1) % NBUF != 0 && CG >= 7, "the epilogue stages 7 x TPP KB in chunk buffer 0 while slow waves may still read the last chunk's buffer");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ab = smem;                                          // [NBUF][CG][TPP][64 lanes][4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* zb = smem + NBUF * CHF + wave * (16 * ZS);
    const int l15 = lane & 15, lq = lane >> 4;
    const int H = g.H, M = g.M, T = g.T;
    const int cg = (int)blockIdx.x % g.ncg, rp = (int)blockIdx.x / g.ncg;     // unit group (16 units), row part
    const int u0 = cg * 16 + wave * 4;                         // this wave's 4 units
    const bool wact = u0 < H;                                  // (the last unit group may be partial: idle waves still move A and keep the barriers)
    const int nwg = gridDim.x;
    const int tb = rp * TPP;                                   // first row tile of this workgroup
    bool tok[TPP];
#pragma unroll
    for (int i = 0; i < TPP; ++i) tok[i] = (tb + i) * 16 < M;

    // ---- this wave's column tile of the recurrent rows -> registers, once: k-step s holds W[kw0 + 4s + lq][column of l15]
    const int ccol = (l15 & 3) * H + u0 + (l15 >> 2);          // W / cinit column of tile column l15 (gate l15 % 4, unit u0 + l15 / 4)
    float breg[4 * NG];
#pragma unroll
    for (int s = 0; s < 4 * NG; ++s) {
        const int k = 4 * s + lq;
        breg[s] = (wact && k < H) ? g.W[(size_t)(g.kw0 + k) * g.ldw + ccol] : 0.0f;
    }

    // ---- the (row, unit) pairs this lane finishes at every step: one per row tile
    const int rt = lane >> 2, uu = lane & 3;
    const int row0 = tb * 16 + rt;
    const int u = u0 + uu;
    float bi = 0.f, bj = 0.f, bf = 0.f, bo = 0.f;
    if (wact) { bi = g.bias[u]; bj = g.bias[H + u]; bf = g.bias[2 * H + u]; bo = g.bias[3 * H + u]; }
    float c_reg[TPP];
    uint32_t vid[TPP], sid[TPP];
#pragma unroll
    for (int i = 0; i < TPP; ++i) {
        const int row = row0 + 16 * i;
        const bool rok = wact && tok[i] && row < M;
        c_reg[i] = (rok && g.c0) ? g.c0[(size_t)row * H + u] : 0.0f;
        vid[i] = (g.keep < 1.0f && rok) ? (uint32_t)g.video_id[row] : 0u;
        sid[i] = (g.keep < 1.0f && rok) ? (uint32_t)g.sample_id[row] : 0u;
    }
    int hoff[TPP], goff[TPP];                                  // byte offsets of this lane's (row, unit) in a [M, H] / [M, 4H] history slot
#pragma unroll
    for (int i = 0; i < TPP; ++i) {
        const int row = row0 + 16 * i;
        const bool rok = wact && tok[i] && row < M;
        hoff[i] = rok ? (row * H + u) * 4 : (int)0x80000000u;
        goff[i] = rok ? (row * 4 * H + u) * 4 : (int)0x80000000u;
    }
    // where this lane's h value of row tile 0 sits in the A-fragment image (tile i: + i * NG * 256 floats)
    const size_t a_own = ((size_t)(tb * NG + (u >> 4)) * 64 + (size_t)((u & 3) * 16 + rt)) * 4 + ((u & 15) >> 2);
    float* const abuf0 = g.abuf;
    float* const abuf1 = g.abuf + (size_t)g.img_tiles * NG * 256;

    GridSync gs{(gu32*)g.sync, g.status, g.fault, g.spin_limit, nwg, false};

    // ---- arrival 0: h_0 in fragment order
#pragma unroll
    for (int i = 0; i < TPP; ++i) {
        const int row = row0 + 16 * i;
        if (wact && tok[i] && row < M) {
            const float h0 = g.h0 ? g.h0[(size_t)row * H + u] : 0.0f;
            __hip_atomic_store((gu32*)(abuf0 + a_own + (size_t)i * NG * 256), __float_as_uint(h0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // the carried partial of a step, branch-free: loop-invariant byte offsets into a per-step buffer resource, an element
    // outside the problem carries an out-of-range offset and reads as zero (20-24 conditional loads per step cost 1.5 us of
    // branches and 64-bit address arithmetic in front of every step)
    float ci[TPP][4];
    int coff[TPP][4];
#pragma unroll
    for (int i = 0; i < TPP; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = (tb + i) * 16 + lq * 4 + r;
            coff[i][r] = (wact && tok[i] && m < M) ? (m * g.ldcinit + ccol) * 4 : (int)0x80000000u;
        }
    auto load_cinit = [&](int t) __attribute__((always_inline)) {
        if (g.cinit && t < g.cinit_steps) {                       // (uniform)
            const __amdgpu_buffer_rsrc_t rsC =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.cinit + (size_t)t * g.cinit_tstride), 0, (int)0x80000000u, 0x00020000);
#pragma unroll
            for (int i = 0; i < TPP; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) ci[i][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsC, coff[i][r], 0, 0));
        } else {
#pragma unroll
            for (int i = 0; i < TPP; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) ci[i][r] = 0.0f;
        }
    };
    load_cinit(0);
    gs.arrive(tid);

    typedef __attribute__((address_space(3))) void* lds_ptr;
#ifdef S2VT_C4_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_readcyclecounter();
#define C4_STAMP(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); st_acc[i] += n_ - st_prev; st_prev = n_; } while (0)
#else
#define C4_STAMP(i) do { } while (0)
#endif
    for (int t = 0; t < T; ++t) {
        f32x4 acc[TPP];
#pragma unroll
        for (int i = 0; i < TPP; ++i) {
            acc[i] = f32x4{ci[i][0], ci[i][1], ci[i][2], ci[i][3]};
            asm volatile("" : "+v"(acc[i]));
        }
#ifndef S2VT_C4_LATE_CINIT
        // the NEXT step's carried partial is requested now: its latency (HBM / Infinity Cache, ~2 us) passes under this step's
        // hand-off wait and MFMAs instead of in front of the next step's (older than every DMA load below: the counted vmcnt
        // waits of the chunk loop still mean what they say)
        if (t + 1 < T) load_cinit(t + 1);
#endif
        C4_STAMP(0);                                              // acc init (waits for the carried partial)
        gs.wait_all((unsigned)t, wave, lane);
        C4_STAMP(1);                                              // grid-wide wait
        // ---- the part's slice of h_t: [TPP row tiles][NG groups] KB, global -> LDS in chunks of CG groups.  Piece p of a chunk
        // = (group p / TPP, row tile p % TPP); wave w issues pieces w, w + 4, ...
        const float* acur = (t & 1) ? abuf1 : abuf0;
        const __amdgpu_buffer_rsrc_t rsA =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(acur + (size_t)tb * NG * 256), 0, TPP * NG * 1024, 0x00020000);
        auto issue_chunk = [&](int c) __attribute__((always_inline)) {
            float* dstb = Ab + (c % NBUF) * CHF;
#pragma unroll
            for (int q = 0; q < DPW; ++q) {
                const int p = wave + 4 * q;
                if (p < TPP * CG) {
                    const int gq = p / TPP, i = p % TPP;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(dstb + (gq * TPP + i) * 256), 16, lane * 16,
                                                             (i * NG + c * CG + gq) * 1024, 0, 16);                     // aux 16 = sc1
                }
            }
        };
        static_assert((TPP * CG) % 4 == 0, "every wave issues exactly DPW loads per chunk (the vmcnt bookkeeping below)");
        static_for<0, NBUF - 1>([&](auto c_) { issue_chunk(decltype(c_)::value); });
        static_for<0, NCH>([&](auto c_) {
            constexpr int c = decltype(c_)::value;
            // my loads of chunk c have landed (those of the later chunks already issued may be in flight); then everybody's
            // have, and everybody is done reading chunk c - 1, whose buffer chunk c + NBUF - 1 now takes
            constexpr int issued = c + NBUF - 1 < NCH ? c + NBUF - 1 : NCH;
            constexpr int later = issued - (c + 1);
            static_assert(later * DPW <= 63, "vmcnt range");
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(later * DPW) : "memory");
            __syncthreads();
            if constexpr (c == 0) C4_STAMP(2);                    // first chunk in LDS
            if constexpr (c + NBUF - 1 < NCH) issue_chunk(c + NBUF - 1);
            const f32x4* ab = reinterpret_cast<const f32x4*>(Ab + (c % NBUF) * CHF) + lane;
            f32x4 a[2][TPP];
#pragma unroll
            for (int i = 0; i < TPP; ++i) a[0][i] = ab[i * 64];
            static_for<0, CG>([&](auto q_) {
                constexpr int gq = decltype(q_)::value;
                if constexpr (gq + 1 < CG) {
#pragma unroll
                    for (int i = 0; i < TPP; ++i) a[(gq + 1) & 1][i] = ab[((gq + 1) * TPP + i) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, 4>([&](auto e_) {
                    constexpr int e = decltype(e_)::value;
                    static_for<0, TPP>([&](auto i_) {
                        constexpr int i = decltype(i_)::value;
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[gq & 1][i][e], breg[(c * CG + gq) * 4 + e], acc[i], 0, 0, 0);
                    });
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        C4_STAMP(3);                                              // chunk loop (MFMAs)
        // ---- per row tile: the gates of a unit meet through the wave's LDS tile; BasicLSTMCell pointwise (EPI_LSTM expressions).
        // Results are STAGED in LDS (the chunk buffers are idle until the next step's first DMA): a wave's tile is 16 rows x
        // 4 units = 16 bytes per row, stored directly that is 140 scattered 4-byte store instructions per workgroup and
        // step (7 history arrays x 5 tiles x 4 waves) whose issue alone cost ~4 us per step; the four waves' units are
        // adjacent, so from LDS one 16-byte store per lane writes 64 contiguous bytes per row (35 instructions), and the
        // hand-off block of a row tile -- exactly ONE 1-KB fragment block, this workgroup's 16 units being one k-group --
        // goes out as one write-through 16-byte store per lane.
        float* const stg = Ab;                                    // [7 arrays][TPP tiles][16 rows][16 units] floats (35-42 KB of the 120)
        constexpr int ST = TPP * 256;                             // floats per staged array
        float hv[TPP], siv[TPP], tjv[TPP], sfv[TPP], sov[TPP];
#pragma unroll
        for (int i = 0; i < TPP; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) zb[(lq * 4 + r) * ZS + l15] = acc[i][r];
            __builtin_amdgcn_wave_barrier();
            const f32x4 z = *reinterpret_cast<const f32x4*>(zb + rt * ZS + uu * 4);
            __builtin_amdgcn_wave_barrier();
            const float zi = z[0] + bi, zj = z[1] + bj, zf = z[2] + bf, zo = z[3] + bo;
            const float si = dm_sigmoidf(zi);
            const float tj = dm_tanhf(zj);
            const float sf = dm_sigmoidf(zf + 1.0f);
            const float so = dm_sigmoidf(zo);
            const float t1 = c_reg[i] * sf;
            const float t2 = si * tj;
            const float cc = t1 + t2;
            const float hvv = dm_tanhf(cc) * so;
            c_reg[i] = cc;
            hv[i] = hvv; siv[i] = si; tjv[i] = tj; sfv[i] = sf; sov[i] = so;
#ifndef S2VT_C4_NOEPI
            stg[1 * ST + (i * 16 + rt) * 16 + wave * 4 + uu] = hvv;       // only what the hand-off needs is staged before it
#endif
        }
        C4_STAMP(4);                                              // pointwise
        __syncthreads();
        if (t + 1 < T) {
            // h_{t+1} blocks: wave w writes row tiles w, w + 4; lane L = kq * 16 + r takes row r, units kq + 4e
            const __amdgpu_buffer_rsrc_t rsN = __builtin_amdgcn_make_buffer_rsrc(((t & 1) ? abuf0 : abuf1) + ((size_t)tb * NG + cg) * 256, 0,
                                                                                   TPP * NG * 1024, 0x00020000);
#pragma unroll
            for (int q = 0; q < (TPP + 3) / 4; ++q) {
                const int i = wave + 4 * q;
                if (i < TPP) {
                    const int r = lane & 15, kq = lane >> 4;
                    u32x4v w4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w4[e] = __float_as_uint(stg[1 * ST + (i * 16 + r) * 16 + kq + 4 * e]);
                    // (rows >= M of a tile and units >= H of the last group carry garbage-free zeros only if they were computed
                    //  from zeros: rows >= M read zero A fragments and zero partials -> finite values, never read back as
                    //  operands of valid rows; units >= H multiply zero weight rows of every consumer)
                    bstore16_sc1(rsN, w4, (i * NG * 256 + lane * 4) * 4, 0);
                }
            }
            gs.arrive(tid);
        }
        C4_STAMP(5);                                              // hand-off stores + drain + arrive
        // ---- behind the hand-off: the DropoutWrapper output (one Philox block per element) and the rest of the staging
#pragma unroll
        for (int i = 0; i < TPP; ++i) {
            float ov = hv[i];
            if (g.out && g.keep < 1.0f)
                ov = (hv[i] / g.keep) * dropout_keep01(g.seed_lo, g.seed_hi, vid[i], sid[i], g.drop_code0 + (uint32_t)t, (uint32_t)u, g.keep);
#ifndef S2VT_C4_NOEPI
            float* sp = stg + (i * 16 + rt) * 16 + wave * 4 + uu;
            sp[0 * ST] = c_reg[i]; sp[2 * ST] = ov; sp[3 * ST] = siv[i]; sp[4 * ST] = tjv[i]; sp[5 * ST] = sfv[i]; sp[6 * ST] = sov[i];
#else
            asm volatile("" ::"v"(ov));
#endif
        }
        __syncthreads();
        // ---- histories (nobody in this launch reads them): block (array, tile) = [16 rows][16 units]; lane = (row, quarter)
        // writes 16 bytes; an element outside the problem carries an out-of-range offset and its store is dropped
        {
            const int srow = lane >> 2, sq = lane & 3;
            const int uq = cg * 16 + sq * 4;
            const __amdgpu_buffer_rsrc_t rsCh = __builtin_amdgcn_make_buffer_rsrc(g.C + (size_t)(t + 1) * g.state_tstride, 0, (int)0x80000000u, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsHh = __builtin_amdgcn_make_buffer_rsrc(g.Hh + (size_t)(t + 1) * g.state_tstride, 0, (int)0x80000000u, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(g.out ? g.out + (size_t)t * g.out_tstride : g.C, 0, g.out ? (int)0x80000000u : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(g.gates ? g.gates + (size_t)t * g.gates_tstride : g.C, 0, g.gates ? (int)0x80000000u : 0, 0x00020000);
            // 7 TPP blocks over the 4 waves: block b = array * TPP + tile
#pragma unroll
            for (int q = 0; q < (7 * TPP + 3) / 4; ++q) {
                const int bidx = wave + 4 * q;
                if (bidx < 7 * TPP) {
                    const int arr = bidx / TPP, i = bidx % TPP;
                    const int row = (tb + i) * 16 + srow;
                    const bool ok = row < M && uq < H;
                    const u32x4v v = __builtin_bit_cast(u32x4v, *reinterpret_cast<const f32x4*>(stg + arr * ST + (i * 16 + srow) * 16 + sq * 4));
                    const int ho = ok ? (row * H + uq) * 4 : (int)0x80000000u;
                    const int go = ok ? (row * 4 * H + uq) * 4 : (int)0x80000000u;
                    if (arr == 0) __builtin_amdgcn_raw_buffer_store_b128(v, rsCh, ho, 0, 0);
                    else if (arr == 1) __builtin_amdgcn_raw_buffer_store_b128(v, rsHh, ho, 0, 0);
                    else if (arr == 2) __builtin_amdgcn_raw_buffer_store_b128(v, rsO, ho, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(v, rsG, go, (arr - 3) * H * 4, 0);
                }
            }
        }
#ifdef S2VT_C4_LATE_CINIT
        if (t + 1 < T) load_cinit(t + 1);
#endif
        C4_STAMP(6);                                              // history stores issued
    }
#ifdef S2VT_C4_STAMP
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 100 || blockIdx.x == 251)) {
        const int slot = (blockIdx.x == 0 ? 0 : blockIdx.x == 100 ? 1 : 2) * 4 + wave;
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(&c4_stamp_acc[slot * 8 + i], st_acc[i]);
    }
#endif
}

// The same with LIVE ROWS (round 4): the rows are VIRTUAL -- row v of the state image is row g.perm[v] of every per-row array
// (carried partial, initial state, noise ids, histories), the caller having sorted the rows by caption length, longest first --
// and step t computes only the row tiles that hold one of the first g.nlive[t] virtual rows: a row behind its caption's <eos>
// feeds nothing (cider_evaluation.py:145-172; the callers gather the live (step, row) pairs for every product outside the
// recurrence), its state simply stops.  The tiles of a part are INTERLEAVED (tile i of part p is image tile 4 i + p), so the live
// prefix spreads evenly over the four parts; the product phase is instantiated per count of live tiles, everything else loops
// over the tiles behind a uniform `i < nl`.  Live rows are computed exactly as in the dense launch: same chains, same pointwise
// expressions, bit-identical histories (tests/test_gpu_live_rows.py).  (A kernel of its own rather than a flag of the one above:
// the dense form's register allocation sits at the limit and does not survive being generated from shared source.)
template <int NG, int TPP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_chain4_live_kernel(const ChainArgs g)
{
    constexpr int ZS = 20;
    constexpr int CG = S2VT_C4_CG;                             // k-groups per chunk
    constexpr int NCH = NG / CG;                               // chunks per step
    constexpr int NBUF = S2VT_C4_NBUF;                         // LDS chunk buffers (NBUF - 1 chunks in flight)
    constexpr int CHF = TPP * CG * 256;                        // floats per chunk buffer
    static_assert(NG % CG == 0 && NCH >= NBUF, "chunking");
    static_assert((NCH - 1) % NBUF != 0 && CG >= 7, "the epilogue stages 7 x TPP KB in chunk buffer 0 while slow waves may still read the last chunk's buffer");
    static_assert(CG % 4 == 0, "every wave issues the same number of DMA loads per chunk for any count of live tiles (the vmcnt bookkeeping below)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ab = smem;                                          // [NBUF][CG][TPP][64 lanes][4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* zb = smem + NBUF * CHF + wave * (16 * ZS);
    int* const arow_l = reinterpret_cast<int*>(smem + NBUF * CHF + 4 * 16 * ZS);   // [TPP][16] rows of the caller's arrays behind this workgroup's virtual rows
    const int l15 = lane & 15, lq = lane >> 4;
    const int H = g.H, M = g.M, T = g.T;
    const int cg = (int)blockIdx.x % g.ncg, rp = (int)blockIdx.x / g.ncg;     // unit group (16 units), row part
    const int u0 = cg * 16 + wave * 4;                         // this wave's 4 units
    const bool wact = u0 < H;                                  // (the last unit group may be partial: idle waves still move A and keep the barriers)
    const int nwg = gridDim.x;
    constexpr int TS = 4;                                      // image tiles between consecutive row tiles of this workgroup
    const int tile0 = rp;                                      // image tile of this workgroup's row tile 0
    bool tok[TPP];
#pragma unroll
    for (int i = 0; i < TPP; ++i) tok[i] = (tile0 + i * TS) * 16 < M;
    auto actual = [&](int v) __attribute__((always_inline)) { return v < M ? (int)g.perm[v] : v; };
    if (tid < TPP * 16) arow_l[tid] = actual((tile0 + (tid >> 4) * TS) * 16 + (tid & 15));
    __syncthreads();

    // ---- this wave's column tile of the recurrent rows -> registers, once: k-step s holds W[kw0 + 4s + lq][column of l15]
    const int ccol = (l15 & 3) * H + u0 + (l15 >> 2);          // W / cinit column of tile column l15 (gate l15 % 4, unit u0 + l15 / 4)
    float breg[4 * NG];
#pragma unroll
    for (int s = 0; s < 4 * NG; ++s) {
        const int k = 4 * s + lq;
        breg[s] = (wact && k < H) ? g.W[(size_t)(g.kw0 + k) * g.ldw + ccol] : 0.0f;
    }

    // ---- the (row, unit) pairs this lane finishes at every step: one per row tile
    const int rt = lane >> 2, uu = lane & 3;
    const int u = u0 + uu;
    float bi = 0.f, bj = 0.f, bf = 0.f, bo = 0.f;
    if (wact) { bi = g.bias[u]; bj = g.bias[H + u]; bf = g.bias[2 * H + u]; bo = g.bias[3 * H + u]; }
    float c_reg[TPP];
    uint32_t vid[TPP], sid[TPP];
    // where this lane's h value of row tile 0 sits in the A-fragment image (tile i: + i * TS * NG * 256 floats)
    const size_t a_own = ((size_t)(tile0 * NG + (u >> 4)) * 64 + (size_t)((u & 3) * 16 + rt)) * 4 + ((u & 15) >> 2);
    float* const abuf0 = g.abuf;
    float* const abuf1 = g.abuf + (size_t)g.img_tiles * NG * 256;
#pragma unroll
    for (int i = 0; i < TPP; ++i) {
        const int vrow = (tile0 + i * TS) * 16 + rt;
        const bool rok = wact && tok[i] && vrow < M;
        const int row = rok ? actual(vrow) : 0;
        c_reg[i] = (rok && g.c0) ? g.c0[(size_t)row * H + u] : 0.0f;
        vid[i] = (g.keep < 1.0f && rok) ? (uint32_t)g.video_id[row] : 0u;
        sid[i] = (g.keep < 1.0f && rok) ? (uint32_t)g.sample_id[row] : 0u;
        // ---- arrival 0: h_0 in fragment order
        if (rok) {
            const float h0 = g.h0 ? g.h0[(size_t)row * H + u] : 0.0f;
            __hip_atomic_store((gu32*)(abuf0 + a_own + (size_t)i * TS * NG * 256), __float_as_uint(h0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    GridSync gs{(gu32*)g.sync, g.status, g.fault, g.spin_limit, nwg, false};

    // the carried partial of a step, branch-free (as above); the offsets go through the row order
    float ci[TPP][4];
    int coff[TPP][4];
#pragma unroll
    for (int i = 0; i < TPP; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = (tile0 + i * TS) * 16 + lq * 4 + r;
            coff[i][r] = (wact && tok[i] && m < M) ? (actual(m) * g.ldcinit + ccol) * 4 : (int)0x80000000u;
        }
    auto load_cinit = [&](int t, int nl) __attribute__((always_inline)) {
        if (g.cinit && t < g.cinit_steps) {                       // (uniform)
            const __amdgpu_buffer_rsrc_t rsC =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.cinit + (size_t)t * g.cinit_tstride), 0, (int)0x80000000u, 0x00020000);
#pragma unroll
            for (int i = 0; i < TPP; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) ci[i][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsC, i < nl ? coff[i][r] : (int)0x80000000u, 0, 0));
        } else {
#pragma unroll
            for (int i = 0; i < TPP; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) ci[i][r] = 0.0f;
        }
    };
    // live tiles of this part at step t: image tiles 0 .. ceil(n / 16) - 1 hold the n live virtual rows; this part owns tiles rp, rp + 4, ..
    auto live_tiles = [&](int t) __attribute__((always_inline)) {
        int n = g.nlive[t];
        n = n < M ? n : M;
        int nl = (((n + 15) >> 4) - rp + 3) >> 2;
        nl = nl < 0 ? 0 : (nl > TPP ? TPP : nl);
        return __builtin_amdgcn_readfirstlane(nl);
    };
    int nl = live_tiles(0);
    load_cinit(0, nl);
    gs.arrive(tid);

    typedef __attribute__((address_space(3))) void* lds_ptr;
    f32x4 acc[TPP];
    // the product phase of a step for NL live row tiles: the part's slice of h_t global -> LDS in chunks, MFMAs (as above)
    auto product = [&](auto nl_, const int t) __attribute__((always_inline)) {
        constexpr int NL = decltype(nl_)::value;
        constexpr int DPW = NL * CG / 4;                          // DMA instructions per wave per chunk
        const float* acur = (t & 1) ? abuf1 : abuf0;
        const __amdgpu_buffer_rsrc_t rsA =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(acur + (size_t)tile0 * NG * 256), 0, ((TPP - 1) * TS + 1) * NG * 1024, 0x00020000);
        auto issue_chunk = [&](int c) __attribute__((always_inline)) {
            float* dstb = Ab + (c % NBUF) * CHF;
#pragma unroll
            for (int q = 0; q < DPW; ++q) {
                const int p = wave + 4 * q;
                const int gq = p / NL, i = p % NL;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(dstb + (gq * TPP + i) * 256), 16, lane * 16,
                                                         (i * TS * NG + c * CG + gq) * 1024, 0, 16);                     // aux 16 = sc1
            }
        };
        static_for<0, NBUF - 1>([&](auto c_) { issue_chunk(decltype(c_)::value); });
        static_for<0, NCH>([&](auto c_) {
            constexpr int c = decltype(c_)::value;
            constexpr int issued = c + NBUF - 1 < NCH ? c + NBUF - 1 : NCH;
            constexpr int later = issued - (c + 1);
            static_assert(later * DPW <= 63, "vmcnt range");
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(later * DPW) : "memory");
            __syncthreads();
            if constexpr (c + NBUF - 1 < NCH) issue_chunk(c + NBUF - 1);
            const f32x4* ab = reinterpret_cast<const f32x4*>(Ab + (c % NBUF) * CHF) + lane;
            f32x4 a[2][NL];
#pragma unroll
            for (int i = 0; i < NL; ++i) a[0][i] = ab[i * 64];
            static_for<0, CG>([&](auto q_) {
                constexpr int gq = decltype(q_)::value;
                if constexpr (gq + 1 < CG) {
#pragma unroll
                    for (int i = 0; i < NL; ++i) a[(gq + 1) & 1][i] = ab[((gq + 1) * TPP + i) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, 4>([&](auto e_) {
                    constexpr int e = decltype(e_)::value;
                    static_for<0, NL>([&](auto i_) {
                        constexpr int i = decltype(i_)::value;
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[gq & 1][i][e], breg[(c * CG + gq) * 4 + e], acc[i], 0, 0, 0);
                    });
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    };

    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int i = 0; i < TPP; ++i) {
            acc[i] = f32x4{ci[i][0], ci[i][1], ci[i][2], ci[i][3]};
            asm volatile("" : "+v"(acc[i]));
        }
        // the NEXT step's carried partial is requested now (older than every DMA load below: the counted vmcnt waits of the chunk
        // loop still mean what they say), for the tiles that will still be live
        const int nl_next = t + 1 < T ? live_tiles(t + 1) : 0;
        if (t + 1 < T) load_cinit(t + 1, nl_next);
        gs.wait_all((unsigned)t, wave, lane);
        if (nl > 0) {
            bool done = false;
            static_for<1, TPP + 1>([&](auto k_) {
                if (!done && nl == decltype(k_)::value) { product(k_, t); done = true; }
            });
        }
        // ---- per row tile: the gates of a unit meet through the wave's LDS tile; BasicLSTMCell pointwise (EPI_LSTM expressions);
        // results staged in LDS (the chunk buffers are idle until the next step's first DMA), as above
        float* const stg = Ab;                                    // [7 arrays][TPP tiles][16 rows][16 units] floats
        constexpr int ST = TPP * 256;                             // floats per staged array
        float hv[TPP], siv[TPP], tjv[TPP], sfv[TPP], sov[TPP];
#pragma unroll
        for (int i = 0; i < TPP; ++i) {
            if (!(i < nl)) continue;                              // (uniform)
#pragma unroll
            for (int r = 0; r < 4; ++r) zb[(lq * 4 + r) * ZS + l15] = acc[i][r];
            __builtin_amdgcn_wave_barrier();
            const f32x4 z = *reinterpret_cast<const f32x4*>(zb + rt * ZS + uu * 4);
            __builtin_amdgcn_wave_barrier();
            const float zi = z[0] + bi, zj = z[1] + bj, zf = z[2] + bf, zo = z[3] + bo;
            const float si = dm_sigmoidf(zi);
            const float tj = dm_tanhf(zj);
            const float sf = dm_sigmoidf(zf + 1.0f);
            const float so = dm_sigmoidf(zo);
            const float t1 = c_reg[i] * sf;
            const float t2 = si * tj;
            const float cc = t1 + t2;
            const float hvv = dm_tanhf(cc) * so;
            c_reg[i] = cc;
            hv[i] = hvv; siv[i] = si; tjv[i] = tj; sfv[i] = sf; sov[i] = so;
            stg[1 * ST + (i * 16 + rt) * 16 + wave * 4 + uu] = hvv;       // only what the hand-off needs is staged before it
        }
        __syncthreads();
        if (t + 1 < T) {
            // h_{t+1} blocks: wave w writes row tiles w, w + 4; lane L = kq * 16 + r takes row r, units kq + 4e
            const __amdgpu_buffer_rsrc_t rsN = __builtin_amdgcn_make_buffer_rsrc(((t & 1) ? abuf0 : abuf1) + ((size_t)tile0 * NG + cg) * 256, 0,
                                                                                   ((TPP - 1) * TS + 1) * NG * 1024, 0x00020000);
#pragma unroll
            for (int q = 0; q < (TPP + 3) / 4; ++q) {
                const int i = wave + 4 * q;
                if (i < nl) {
                    const int r = lane & 15, kq = lane >> 4;
                    u32x4v w4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w4[e] = __float_as_uint(stg[1 * ST + (i * 16 + r) * 16 + kq + 4 * e]);
                    bstore16_sc1(rsN, w4, (i * TS * NG * 256 + lane * 4) * 4, 0);
                }
            }
            gs.arrive(tid);
        }
        // ---- behind the hand-off: the DropoutWrapper output (one Philox block per element) and the rest of the staging
#pragma unroll
        for (int i = 0; i < TPP; ++i) {
            if (!(i < nl)) continue;
            float ov = hv[i];
            if (g.out && g.keep < 1.0f)
                ov = (hv[i] / g.keep) * dropout_keep01(g.seed_lo, g.seed_hi, vid[i], sid[i], g.drop_code0 + (uint32_t)t, (uint32_t)u, g.keep);
            float* sp = stg + (i * 16 + rt) * 16 + wave * 4 + uu;
            sp[0 * ST] = c_reg[i]; sp[2 * ST] = ov; sp[3 * ST] = siv[i]; sp[4 * ST] = tjv[i]; sp[5 * ST] = sfv[i]; sp[6 * ST] = sov[i];
        }
        __syncthreads();
        // ---- histories (nobody in this launch reads them): block b = tile * 7 + array = [16 rows][16 units]; lane = (row, quarter)
        // writes 16 bytes to the row the virtual row stands for; an element outside the problem carries an out-of-range offset
        if (nl > 0) {
            const int srow = lane >> 2, sq = lane & 3;
            const int uq = cg * 16 + sq * 4;
            const __amdgpu_buffer_rsrc_t rsCh = __builtin_amdgcn_make_buffer_rsrc(g.C + (size_t)(t + 1) * g.state_tstride, 0, (int)0x80000000u, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsHh = __builtin_amdgcn_make_buffer_rsrc(g.Hh + (size_t)(t + 1) * g.state_tstride, 0, (int)0x80000000u, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(g.out ? g.out + (size_t)t * g.out_tstride : g.C, 0, g.out ? (int)0x80000000u : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(g.gates ? g.gates + (size_t)t * g.gates_tstride : g.C, 0, g.gates ? (int)0x80000000u : 0, 0x00020000);
#pragma unroll
            for (int q = 0; q < (7 * TPP + 3) / 4; ++q) {
                const int bidx = wave + 4 * q;
                const int i = bidx / 7, arr = bidx - 7 * i;
                if (i < nl) {
                    const int vrow = (tile0 + i * TS) * 16 + srow;
                    const bool ok = vrow < M && uq < H;
                    const int row = arow_l[i * 16 + srow];
                    const u32x4v v = __builtin_bit_cast(u32x4v, *reinterpret_cast<const f32x4*>(stg + arr * ST + (i * 16 + srow) * 16 + sq * 4));
                    const int ho = ok ? (row * H + uq) * 4 : (int)0x80000000u;
                    const int go = ok ? (row * 4 * H + uq) * 4 : (int)0x80000000u;
                    if (arr == 0) __builtin_amdgcn_raw_buffer_store_b128(v, rsCh, ho, 0, 0);
                    else if (arr == 1) __builtin_amdgcn_raw_buffer_store_b128(v, rsHh, ho, 0, 0);
                    else if (arr == 2) __builtin_amdgcn_raw_buffer_store_b128(v, rsO, ho, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(v, rsG, go, (arr - 3) * H * 4, 0);
                }
            }
        }
        nl = nl_next;
    }
}

typedef void (*ChainFn)(const ChainArgs);
struct ChainCfg { int ng, tmw, nc; ChainFn fn; const char* name; ChainFn fn_live; const char* name_live; };     // nc == 4: the register-weights form (tmw = row tiles per part); fn_live: its live-row variant
constexpr int kMaxTmw = 6;                                     // 6 row tiles per wave x 4 waves x 16 rows = 384 rows
const ChainCfg kChain[] = {
    {8, 1, 1, lstm_chain_kernel<8, 1, 1>, "chain(ng8,m64)"},      {8, 2, 1, lstm_chain_kernel<8, 2, 1>, "chain(ng8,m128)"},
    {8, 3, 1, lstm_chain_kernel<8, 3, 1>, "chain(ng8,m192)"},     {8, 4, 1, lstm_chain_kernel<8, 4, 1>, "chain(ng8,m256)"},
    {8, 5, 1, lstm_chain_kernel<8, 5, 1>, "chain(ng8,m320)"},     {8, 6, 1, lstm_chain_kernel<8, 6, 1>, "chain(ng8,m384)"},
    {64, 1, 1, lstm_chain_kernel<64, 1, 1>, "chain(ng64,m64)"},   {64, 2, 1, lstm_chain_kernel<64, 2, 1>, "chain(ng64,m128)"},
    {64, 3, 1, lstm_chain_kernel<64, 3, 1>, "chain(ng64,m192)"},  {64, 4, 1, lstm_chain_kernel<64, 4, 1>, "chain(ng64,m256)"},
    {64, 5, 1, lstm_chain_kernel<64, 5, 1>, "chain(ng64,m320)"},  {64, 6, 1, lstm_chain_kernel<64, 6, 1>, "chain(ng64,m384)"},
    // 8 units x half the row tiles per workgroup (M > 64): tmw = row tiles per wave of a part
    {64, 1, 2, lstm_chain_kernel<64, 1, 2>, "chain2(ng64,m128)"}, {64, 2, 2, lstm_chain_kernel<64, 2, 2>, "chain2(ng64,m256)"},
    {64, 3, 2, lstm_chain_kernel<64, 3, 2>, "chain2(ng64,m384)"},
    // weights in registers, 16 units x a quarter of the row tiles per workgroup (M > 256): tmw = row tiles per part
    {64, 5, 4, lstm_chain4_kernel<64, 5>, "chain4(ng64,m320)", lstm_chain4_live_kernel<64, 5>, "chain4(ng64,m320)[live]"},
    {64, 6, 4, lstm_chain4_kernel<64, 6>, "chain4(ng64,m384)", lstm_chain4_live_kernel<64, 6>, "chain4(ng64,m384)[live]"},
};
constexpr int kNumCfg = (int)(sizeof(kChain) / sizeof(kChain[0]));
constexpr int kMaxDev = 32;
// one-time state PER DEVICE (a process may drive several): CU count, the dynamic-LDS attribute of every configuration, how
// many workgroups of each configuration one CU holds, the device-resident fault word
struct DevState {
    std::once_flag once;
    int num_cus = 0;                       // 0 = the persistent form is unavailable on this device
    int per_cu[kNumCfg] = {};
    unsigned* fault = nullptr;             // device memory, 1 after a timed-out wait until chain_ack()
};
DevState g_dev[kMaxDev];
std::mutex g_launch_mu;
hipEvent_t g_last_done = nullptr;          // recorded behind the most recent persistent launch of this process
hipStream_t g_last_stream = nullptr;
int g_last_device = -1;
std::atomic<unsigned> g_acked{0};          // timeouts acknowledged so far (chain_ack)
std::atomic<bool> g_disabled{false};       // chain_ack(disable): per-step launches from here on
std::atomic<int> g_hold{0};                // chain_hold(): per-step launches while > 0 (a collective's kernels are in flight beside us)

int chain_lds_bytes(const ChainCfg& c)
{
    if (c.nc == 4) return (S2VT_C4_NBUF * c.tmw * S2VT_C4_CG * 256 + 4 * 16 * 20 + c.tmw * 16) * 4;     // chunk buffers of tmw x CG KB-pieces + the per-wave z tiles + (live form) the rows behind the virtual rows
    return (c.ng * c.nc * 256 + 4 * 16 * 20) * 4;
}

int current_num_cus()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return 0;
    return g_dev[dev].num_cus;
}
bool chain_two_parts(int M, int H)                             // the 8-unit / half-the-rows form serves this shape
{
    static const bool off = [] { const char* e = getenv("S2VT_CHAIN2"); return e && e[0] == '0'; }();          // dev knob
    return !off && M > 64 && (H + 15) / 16 > 8 && (H & 7) == 0 && 2 * (H / 8) <= current_num_cus();
}
bool chain_four_parts(int M, int H)                            // the register-weights form serves this shape
{
    static const bool on = [] { const char* e = getenv("S2VT_CHAIN4"); return !(e && e[0] == '0'); }();        // dev knob
    return on && M > 256 && (H + 15) / 16 > 8 && (H & 3) == 0 && 4 * ((H + 15) / 16) <= current_num_cus();
}
int chain_cfg(int M, int H)                                    // index into kChain, or -1
{
    const int tiles = (M + 15) / 16;
    if (chain_four_parts(M, H)) {
        const int tpp = (tiles + 3) / 4;
        for (int i = 0; i < kNumCfg; ++i)
            if (kChain[i].nc == 4 && kChain[i].ng == 64 && kChain[i].tmw == (tpp <= 5 ? 5 : 6)) return i;
        return -1;
    }
    const bool two = chain_two_parts(M, H);
    const int ng = (H + 15) / 16 <= 8 ? 8 : 64, tmw = two ? ((tiles + 1) / 2 + 3) / 4 : (M + 63) / 64, nc = two ? 2 : 1;
    for (int i = 0; i < kNumCfg; ++i)
        if (kChain[i].ng == ng && kChain[i].tmw == tmw && kChain[i].nc == nc) return i;
    return -1;
}

std::once_flag g_status_once;
unsigned* g_status_host = nullptr;     // pinned, device-mapped (portable): timeouts of every chain launch of this process
unsigned* g_status_dev = nullptr;

DevState* dev_state()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
    std::call_once(g_status_once, [] {
        void* hp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess) {
            g_status_host = static_cast<unsigned*>(hp);
            *g_status_host = 0u;
            void* dp = nullptr;
            if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) g_status_dev = static_cast<unsigned*>(dp);
        }
    });
    DevState& d = g_dev[dev];
    std::call_once(d.once, [&d, dev] {
        hipDeviceProp_t p;
        int cus = 0;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        bool ok = cus > 0 && g_status_dev;
        for (int i = 0; ok && i < kNumCfg; ++i) {
            const ChainCfg& c = kChain[i];
            ok = hipFuncSetAttribute(reinterpret_cast<const void*>(c.fn), hipFuncAttributeMaxDynamicSharedMemorySize, chain_lds_bytes(c)) == hipSuccess;
            if (ok && c.fn_live)
                ok = hipFuncSetAttribute(reinterpret_cast<const void*>(c.fn_live), hipFuncAttributeMaxDynamicSharedMemorySize, chain_lds_bytes(c)) == hipSuccess;
            int n = 0;
            // co-residency is CHECKED, not assumed: workgroups of this configuration one CU can hold (registers, LDS, waves)
            if (ok && hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(c.fn), 256, chain_lds_bytes(c)) == hipSuccess)
                d.per_cu[i] = n;
        }
        void* f = nullptr;
        if (ok && hipMalloc(&f, 256) == hipSuccess && hipMemset(f, 0, 256) == hipSuccess) d.fault = static_cast<unsigned*>(f);
        d.num_cus = (ok && d.fault) ? cus : 0;                // an LDS request refused / no status word: the per-step path serves
    });
    return &d;
}


}  // namespace

size_t chain_scratch_floats(int H)
{
    const int ng = (H + 15) / 16 <= 8 ? 8 : 64;
    return (size_t)2 * 4 * kMaxTmw * ng * 256;                // two fragment images of h (ping-pong), up to 384 rows
}

static int chain_max_rows()
{
    static const int v = [] { const char* e = getenv("S2VT_CHAIN_MAXM"); return e ? atoi(e) : 64 * kMaxTmw; }();   // dev knob
    return v;
}

bool chain_eligible(int M, int H)
{
    static const bool off = [] { const char* e = getenv("S2VT_CHAIN"); return e && e[0] == '0'; }();
    if (off || chain_persistent_disabled()) return false;
    DevState* d = dev_state();
    if (!d || d->num_cus <= 0) return false;
    if (!(M >= 1 && M <= chain_max_rows() && H >= 4 && (H & 3) == 0 && H <= 1024 && H / 4 <= d->num_cus)) return false;
    const int ci = chain_cfg(M, H);
    if (ci < 0) return false;
    const int grid = kChain[ci].nc == 4 ? 4 * ((H + 15) / 16) : kChain[ci].nc * (H / (4 * kChain[ci].nc));
    return (long)d->per_cu[ci] * d->num_cus >= grid;           // every workgroup of the grid fits on the chip at once
}

bool bwd_chain_live_capable(int M, int H);                     // (chain_bwd.hip)
bool chain_live_capable(int M, int H)
{
    static const bool off = [] { const char* e = getenv("S2VT_CHAIN_LIVE"); return e && e[0] == '0'; }();      // dev knob: dense recurrences under live-row updates
    if (off || !chain_eligible(M, H)) return false;
    const int ci = chain_cfg(M, H);
    return ci >= 0 && kChain[ci].fn_live != nullptr && bwd_chain_live_capable(M, H);
}

static unsigned spin_limit()
{
    static const unsigned v = [] { const char* e = getenv("S2VT_CHAIN_SPIN_LIMIT"); return e ? (unsigned)strtoul(e, nullptr, 10) : kSpinLimitDefault; }();   // dev / test knob
    return v;
}

hipError_t launch_lstm_chain(const ChainArgs& a, hipStream_t st)
{
    if (!chain_eligible(a.M, a.H)) return hipErrorInvalidValue;
    if (a.T <= 0) return hipSuccess;
    if (!chain_operands_ok(a.W, a.ldw, a.abuf)) return hipErrorInvalidValue;
    DevState* d = dev_state();
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int ci = chain_cfg(a.M, a.H);
    const ChainCfg& c = kChain[ci];
    ChainArgs a2 = a;
    const bool live = a.perm && a.nlive && c.fn_live;          // (other forms: a dense launch, the same results)
    if (!live) { a2.perm = nullptr; a2.nlive = nullptr; }
    const ChainFn fn = live ? c.fn_live : c.fn;
    a2.status = g_status_dev;
    a2.fault = d->fault;
    a2.spin_limit = spin_limit();
    const int tiles = (a.M + 15) / 16, parts = c.nc;           // (two column tiles go with two row parts; nc == 4: four parts of tmw row tiles)
    if (c.nc == 4) {
        a2.ncg = (a.H + 15) / 16;
        a2.tpp = c.tmw;
        a2.img_tiles = 4 * c.tmw;
    } else {
        a2.ncg = a.H / (4 * c.nc);
        a2.tpp = parts == 1 ? 4 * c.tmw : (tiles + 1) / 2;
        a2.img_tiles = parts * 4 * c.tmw;
    }
    // One persistent grid at a time per process: a launch on a different stream (or device) waits for the previous one.
    std::lock_guard<std::mutex> lk(g_launch_mu);
    if (g_last_done && (g_last_stream != st || g_last_device != dev)) {
        hipError_t we = hipStreamWaitEvent(st, g_last_done, 0);
        if (we != hipSuccess) return we;
    }
    // every polled word and the fragment images start from zero on EVERY call (a memset node ahead of the launch)
    ZeroList z;
    z.add(a.sync, kChainSyncBytes); z.add(a.abuf, (size_t)2 * a2.img_tiles * c.ng * 256 * 4);
    hipError_t e = launch_zero_regions(z, st);
    if (e != hipSuccess) return e;
    const int lds = chain_lds_bytes(c);
    const dim3 grid((unsigned)(parts * a2.ncg));
    const double flops = 2.0 * a.M * (double)a.H * 4.0 * a.H * a.T;
    auto mark_done = [&]() -> hipError_t {
        if (!g_last_done || g_last_device != dev) {            // (an event belongs to the device it was created on)
            if (g_last_done) (void)hipEventDestroy(g_last_done);
            g_last_done = nullptr;
            hipError_t ce = hipEventCreateWithFlags(&g_last_done, hipEventDisableTiming);
            if (ce != hipSuccess) return ce;
        }
        g_last_stream = st;
        g_last_device = dev;
        return hipEventRecord(g_last_done, st);
    };
    const int pci = live ? ci + 100 : ci;                      // (the live-row variant is a profiler row of its own)
    if (!prof_wants(5, pci)) {
        hipLaunchKernelGGL(fn, grid, dim3(256), lds, st, a2);
        e = hipGetLastError();
        return e != hipSuccess ? e : mark_done();
    }
    hipEvent_t e0, e1;
    hipError_t pe = prof_events(&e0, &e1);
    if (pe != hipSuccess) return pe;
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL(fn, grid, dim3(256), lds, st, a2);
    (void)hipEventRecord(e1, st);
    prof_record(5, pci, live ? c.name_live : c.name, live ? 0.0 : flops, e0, e1);     // (live rows: the executed count lives on the device -- no rate is claimed)
    e = hipGetLastError();
    return e != hipSuccess ? e : mark_done();
}

bool chain_operands_ok(const float* W, int ldw, const float* abuf)
{
    return !(reinterpret_cast<uintptr_t>(W) & 15) && !(ldw & 3) && !(reinterpret_cast<uintptr_t>(abuf) & 15);
}


// ---- gated overlap (internal.h ChainGate)
namespace {
__global__ void chain_gate_kernel(const unsigned* sync, unsigned arrivals, unsigned limit)
{
    if (threadIdx.x != 0) return;
    for (unsigned spins = 0; spins < limit; ++spins) {
        unsigned sum = 0;
        for (int s = 0; s < kShards; ++s) sum += __hip_atomic_load((const gu32*)(sync + s * 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sum >= arrivals) break;                            // every workgroup of the grid has run up to its first hand-off: all resident
        __builtin_amdgcn_s_sleep(8);
    }                                                          // (a grid that never arrives -- a failed launch -- costs ~10 ms here, nothing else)
}
thread_local ChainGate* t_gate = nullptr;
}  // namespace
void chain_gate_arm(ChainGate* g) { t_gate = g; if (g) g->fired = false; }
hipError_t chain_gate_zeroed(hipStream_t st) { return t_gate ? hipEventRecord(t_gate->ev, st) : hipSuccess; }
hipError_t chain_gate_launched(const unsigned* sync, unsigned arrivals)
{
    ChainGate* g = t_gate;
    if (!g) return hipSuccess;
    t_gate = nullptr;
    hipError_t e = hipStreamWaitEvent(g->side, g->ev, 0);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(chain_gate_kernel, dim3(1), dim3(64), 0, g->side, sync, arrivals, 1u << 15);
    g->fired = true;
    return hipGetLastError();
}

// ---- the launch state chain_bwd.hip shares (chain_common.h)
bool chain_host(ChainHost* out)
{
    DevState* d = dev_state();
    int dev = 0;
    if (!d || hipGetDevice(&dev) != hipSuccess) return false;
    out->num_cus = d->num_cus; out->device = dev; out->status_dev = g_status_dev; out->fault = d->fault; out->spin_limit = spin_limit();
    return d->num_cus > 0;
}
bool chain_persistent_disabled() { return g_disabled.load(std::memory_order_relaxed) || g_hold.load(std::memory_order_relaxed) > 0; }
void chain_hold(bool on) { if (on) g_hold.fetch_add(1, std::memory_order_relaxed); else if (g_hold.load(std::memory_order_relaxed) > 0) g_hold.fetch_sub(1, std::memory_order_relaxed); }
ChainLaunchOrder::ChainLaunchOrder() { g_launch_mu.lock(); }
ChainLaunchOrder::~ChainLaunchOrder() { g_launch_mu.unlock(); }
hipError_t ChainLaunchOrder::before(hipStream_t st, int dev)
{
    if (g_last_done && (g_last_stream != st || g_last_device != dev)) return hipStreamWaitEvent(st, g_last_done, 0);
    return hipSuccess;
}
hipError_t ChainLaunchOrder::after(hipStream_t st, int dev)
{
    if (!g_last_done || g_last_device != dev) {            // (an event belongs to the device it was created on)
        if (g_last_done) (void)hipEventDestroy(g_last_done);
        g_last_done = nullptr;
        hipError_t ce = hipEventCreateWithFlags(&g_last_done, hipEventDisableTiming);
        if (ce != hipSuccess) return ce;
    }
    g_last_stream = st;
    g_last_device = dev;
    return hipEventRecord(g_last_done, st);
}

unsigned chain_timeouts() { return g_status_host ? *static_cast<volatile unsigned*>(g_status_host) : 0u; }

bool chain_fault() { return chain_timeouts() != g_acked.load(std::memory_order_relaxed); }

const unsigned* chain_fault_word()
{
    DevState* d = dev_state();
    return d ? d->fault : nullptr;
}

hipError_t chain_ack(bool disable)
{
    // the caller has synchronised the device (s2vt_chain_ack does): no kernel is reading or raising the words now
    g_acked.store(chain_timeouts(), std::memory_order_relaxed);
    if (disable) g_disabled.store(true, std::memory_order_relaxed);
    for (int i = 0; i < kMaxDev; ++i)
        if (g_dev[i].fault) {
            int cur = 0;
            (void)hipGetDevice(&cur);
            (void)hipSetDevice(i);
            const hipError_t e = hipMemset(g_dev[i].fault, 0, 256);
            (void)hipSetDevice(cur);
            if (e != hipSuccess) return e;
        }
    return hipSuccess;
}

}  // namespace s2vt

#ifdef S2VT_C4_STAMP
// dev build only: per-phase cycle sums of lstm_chain4_kernel (3 workgroups x 4 waves x 8 phases), read and reset
extern "C" int s2vt_c4_stamp_read(unsigned long long* out96)
{
    if (hipDeviceSynchronize() != hipSuccess) return -4;
    if (hipMemcpyFromSymbol(out96, HIP_SYMBOL(s2vt::c4_stamp_acc), 96 * sizeof(unsigned long long)) != hipSuccess) return -4;
    unsigned long long z[96] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(s2vt::c4_stamp_acc), z, sizeof(z)) == hipSuccess ? 0 : -4;
}
#endif
