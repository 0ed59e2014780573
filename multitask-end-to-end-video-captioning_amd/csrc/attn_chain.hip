// attn_chain.hip -- the whole recurrence of the temporal-attention captioner's unroll (original_attention.py:109-135) in ONE
// persistent launch: per decode step  query projection -> score / softmax / context -> LSTM3, T steps, B <= 64 rows.
//
// As separate launches a step costs ~65 us (hWa = out @ Wa 23 us, attention 17 us, the LSTM3 step 25 us) for 0.4 GFLOP, all
// of it latency: every launch re-streams its weights from L2, refills and drains.  Here nothing but the state moves
// (the construction of chain.hip, extended by two more roles and two more hand-offs per step):
//   * every workgroup j (grid = H / 4, one per CU) owns 4 hidden units = 16 gate columns of LSTM3.  Its slice of the
//     RECURRENT rows W3[2H:3H] sits in LDS in MFMA B-fragment order (64 KB), its slice of the CONTEXT rows W3[0:H] in
//     registers (4 NG floats per lane, the same in all four waves: at one wave per SIMD a wave has 512 VGPRs); wave w takes
//     row tile w.  The chain of a pre-activation is the contract's: hoisted embedding partial ->
//     h rows -> context rows, each block in ascending k on v_mfma_f32_16x16x4_f32.
//   * QUERY role (workgroups 0 .. ceil(H/16) - 1): one 16-column tile of Wa in LDS (64 KB); at the start of a step the four
//     waves run out_{t-1} @ Wa for their row tiles (A fragments straight from the fragment-order image of the dropped
//     output) and publish hWa_t row-major (write-through): it is also the history the backward reads.
//   * ATTENTION role (the B workgroups from the top of the grid, one batch row each): waits for hWa_t, tanh(hWa + P) for up
//     to 5 frames at a time into LDS, the score chains on one VALU lane per frame (the arithmetic of attn.hip's
//     attn_fwd_kernel, bit for bit), softmax over the frames, context -> the row-major history AND the fragment-order image
//     every workgroup's context block reads.
//   * three hand-offs per step, the measured sc1 form (MI355X_MICROARCH.md "Valid forms", row 1): end of step (h_t and
//     out_t images; all workgroups, sharded counter), hWa_t (query workgroups -> attention workgroups), ctx_t (attention
//     workgroups -> all).  While the query / attention roles work, everybody runs the h block of its chain, which needs
//     nothing of the current step.
// Critical path per step ~ hWa chain (4 NG dependent MFMAs) + attention (~4 us) + context block (4 NG MFMAs) + pointwise +
// three hand-offs, against ~65 us as launches.  States, gates, dropped outputs, hWa, alphas and contexts are bit-identical to
// the per-step launches (tests/test_gpu_attention_model.py runs both forms).  Co-residency, bounded spins, the sticky fault
// and the one-persistent-grid-at-a-time ordering are chain.hip's (chain_common.h).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "attn_score.h"
#include "chain_common.h"

namespace s2vt {

namespace {

constexpr int kACRows = 5;                 // frames whose tanh rows are resident in LDS at a time
constexpr int kACMaxTv = 64;

struct AttnChainKArgs {
    const float* W3; int ldw;              // [3H, 4H]: rows [0,H) context, [H,2H) embedding (hoisted by the caller), [2H,3H) h
    const float* b3;                       // [4H]
    const float* cinit; size_t cinit_tstride; int ldcinit;    // carried partial of step t (the hoisted embedding block), rows ldcinit apart
    float* C; float* Hh; float* Out; size_t state_tstride;    // histories: step t writes slot t + 1
    float* gates; size_t gates_tstride;    // [T][B][4H] activated gates (may alias cinit: read before written)
    const float* Wa; int ldwa;             // [H, H]
    const float* P; const float* Vt;       // [Tv, B, H]
    const float* w;                        // [H]
    float* hWa; size_t hwa_tstride;        // [T][B][H]; slot t is written for t >= 1 (step 0's query is the zero state)
    float* alpha; float* asum; float* ctx; // [T][Tv][B], [T][B], [T][B][H]
    int B, H, T, Tv;
    float keep; uint32_t seed_lo, seed_hi, drop_code0;
    const int32_t* video_id; const int32_t* sample_id;
    float* himg; float* qimg; float* cimg; // fragment images: h and dropped-out double-buffered [2][4][NG][256] (qimg == himg when keep >= 1), ctx [4][NG][256]
    unsigned* sync;                        // three sharded counter sets (end of step, hWa, ctx): 3 x kChainSyncBytes
    unsigned* status; unsigned* fault; unsigned spin_limit;
};

#ifdef S2VT_AC_STAMP
__device__ unsigned long long ac_stamp_acc[3 * 16];          // dev build: per-phase clock sums of three workgroups (query role, plain, attention role)
#endif
template <int NG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_chain_kernel(const AttnChainKArgs g)
{
    constexpr int ZS = 20;
    constexpr int RING = NG < 16 ? NG : 16;
    constexpr int KP = NG * 16;                                // padded reduction length
    constexpr int LDT = KP + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                                          // [NG][64][4]  B fragments, W3 rows [2H, 3H)
    float* Wq = smem + NG * 256 + 4 * 16 * ZS;                 // [NG][64][4]  B fragments, Wa column tile (query role)
    const int tid = threadIdx.x, lane = tid & 63;
    const int pwave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* zb = smem + NG * 256 + pwave * (16 * ZS);
    float* Tq = smem + 2 * NG * 256 + 4 * 16 * ZS;             // [kACRows][LDT] tanh rows (attention role) behind the query tile ...
    float* wl = Tq + kACRows * LDT;                            // [KP] the score vector
    float* ev = wl + KP;                                       // [64] scores, then alphas
    float* xv = ev + kACMaxTv;                                 // [64] exp(e)
    float* sc = xv + kACMaxTv;                                 // [4]
    const int wave = (pwave + (int)blockIdx.x) & 3;            // this wave's row tile (rotated per workgroup: spreads the image lines over L2 channels)
    const int l15 = lane & 15, lq = lane >> 4;
    const int H = g.H, M = g.B, T = g.T, Tv = g.Tv;
    const int nwg = gridDim.x;
    const int u0 = (int)blockIdx.x * 4;
    const int nq = (H + 15) >> 4;
    const bool roleQ = (int)blockIdx.x < nq;
    const int brow = nwg - 1 - (int)blockIdx.x;                // attention role: batch row
    const bool roleA = brow < M;
    const bool tok = wave * 16 < M;
    // ... and a workgroup that is NOT a query workgroup has no use for that tile's 64 KB: its tanh rows start there, and a chunk holds
    // (NG * 256 + kACRows * LDT) / LDT frames instead of kACRows (20 instead of 5 at NG = 64) -- the score chains of a chunk run side by
    // side on one lane each, so a chunk costs about the same whatever it holds: Tv = 32 in 2 chunks instead of 7
    float* const Tt = roleQ ? Tq : Wq;
    const int RCA = roleQ ? kACRows : (NG * 256 + kACRows * LDT) / LDT;

    // ---- prologue: weights to their places, once
    for (int idx = tid; idx < NG * 16 * 4; idx += 256) {       // recurrent rows of W3 -> LDS (chain.hip's layout)
        const int k = idx >> 2, gt = idx & 3;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < H) v = *reinterpret_cast<const f32x4*>(g.W3 + (size_t)(2 * H + k) * g.ldw + (size_t)gt * H + u0);
        const int j = k >> 4, e = (k & 15) >> 2, kq = k & 3;
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) Wl[((j * 64 + kq * 16 + uu * 4 + gt) << 2) + e] = v[uu];
    }
    if (roleQ) {                                                // Wa[:, 16 j .. 16 j + 15] -> LDS, column cc of the tile = lane % 16
        const int c0 = (int)blockIdx.x * 16;
        for (int idx = tid; idx < NG * 16 * 4; idx += 256) {
            const int k = idx >> 2, c4 = idx & 3;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < H && c0 + 4 * c4 < H) v = *reinterpret_cast<const f32x4*>(g.Wa + (size_t)k * g.ldwa + c0 + 4 * c4);
            const int j = k >> 4, e = (k & 15) >> 2, kq = k & 3;
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) Wq[((j * 64 + kq * 16 + c4 * 4 + uu) << 2) + e] = v[uu];
        }
    }
    if (roleA) {
        for (int h = tid; h < KP; h += 256) wl[h] = h < H ? g.w[h] : 0.f;
        for (int i = tid; i < RCA * (KP - H); i += 256) Tt[(size_t)(i / (KP - H)) * LDT + H + i % (KP - H)] = 0.f;     // pad columns: zeros for good
    }
    // context rows W3[0:H] of this workgroup's 16 gate columns -> registers: k-step s of group j holds W3[16 j + 4 e + lq][column of l15]
    const int ccol = (l15 & 3) * H + u0 + (l15 >> 2);          // W / cinit column of tile column l15 (gate l15 % 4, unit u0 + l15 / 4)
    float breg[4 * NG];
#pragma unroll
    for (int s = 0; s < 4 * NG; ++s) {
        const int k = 4 * s + lq;
        breg[s] = k < H ? g.W3[(size_t)k * g.ldw + ccol] : 0.0f;
    }
    const int rt = lane >> 2, uu = lane & 3;                   // the (row, unit) pair this lane finishes: row rt of its tile, unit u0 + uu
    const int row = wave * 16 + rt;
    const bool rok = tok && row < M;
    const int u = u0 + uu;
    const float bi = g.b3[u], bj = g.b3[H + u], bf = g.b3[2 * H + u], bo = g.b3[3 * H + u];
    float c_reg = 0.0f;                                        // zero initial state (:100-101)
    const uint32_t vid = (g.keep < 1.0f && rok) ? (uint32_t)g.video_id[row] : 0u;
    const uint32_t sid = (g.keep < 1.0f && rok) ? (uint32_t)g.sample_id[row] : 0u;
    // where this lane's value for (row, unit u) sits in a fragment image: k = u -> group u / 16, lane (u % 4) * 16 + rt, component (u % 16) / 4
    const size_t a_own = ((size_t)(wave * NG + (u >> 4)) * 64 + (size_t)((u & 3) * 16 + rt)) * 4 + ((u & 15) >> 2);
    const size_t img_floats = (size_t)4 * NG * 256;
    const bool drops = g.keep < 1.0f;

    GridSync gs{(gu32*)g.sync, g.status, g.fault, g.spin_limit, nwg, false};
    // the two role hand-offs have their own SHARDED counters as well: with one word each, ~250 pollers hammered the L2 line the
    // 63 / 64 arrivals had to get through, and a hand-off took ~6 us instead of ~2 (s_memtime stamps)
    GridSync gq{(gu32*)g.sync + kChainSyncBytes / 4, g.status, g.fault, g.spin_limit, nq, false};        // hWa: the query workgroups arrive
    GridSync gc{(gu32*)g.sync + 2 * (kChainSyncBytes / 4), g.status, g.fault, g.spin_limit, M, false};    // ctx: the attention workgroups arrive

    float ci[4], cn[4];
    auto load_cinit = [&](int t, float* dst) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = wave * 16 + lq * 4 + r;
            dst[r] = (g.cinit && tok && m < M) ? g.cinit[(size_t)t * g.cinit_tstride + (size_t)m * g.ldcinit + ccol] : 0.0f;
        }
    };
    load_cinit(0, ci);
    const int voff = tok ? lane * 16 : (int)0x80000000u;       // a wave without rows reads zeros (out of range)

    // one row tile x 16 columns over the whole reduction, A fragments from a fragment image (sc1 loads into a register ring),
    // B fragments from LDS (one ds_read_b128 = four k-steps)
    auto pass_lds = [&](const float* tile, const float* Wb, f32x4 acc) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tile), 0, NG * 1024, 0x00020000);
        f32x4 a[RING];
        static_for<0, RING>([&](auto j_) { constexpr int j = decltype(j_)::value; a[j] = bload16_sc1(rsA, voff, j * 1024); });
        __builtin_amdgcn_sched_barrier(0);
        const f32x4* bl = reinterpret_cast<const f32x4*>(Wb) + lane;
        constexpr int PB = NG < 4 ? NG : 4;
        f32x4 b[PB];
        static_for<0, PB>([&](auto j_) { constexpr int j = decltype(j_)::value; b[j] = bl[j * 64]; });
        static_for<0, NG>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            const f32x4 bj4 = b[j % PB];
            if constexpr (j + PB < NG) b[j % PB] = bl[(j + PB) * 64];
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, 4>([&](auto e_) {
                constexpr int e = decltype(e_)::value;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j % RING][e], bj4[e], acc, 0, 0, 0);
            });
            if constexpr (j + RING < NG) {
                __builtin_amdgcn_sched_barrier(0);
                a[j % RING] = bload16_sc1(rsA, voff, (j + RING) * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        return acc;
    };
    // the same with the B fragments in registers (the context block)
    auto pass_reg = [&](const float* tile, f32x4 acc) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tile), 0, NG * 1024, 0x00020000);
        f32x4 a[RING];
        static_for<0, RING>([&](auto j_) { constexpr int j = decltype(j_)::value; a[j] = bload16_sc1(rsA, voff, j * 1024); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, NG>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            static_for<0, 4>([&](auto e_) {
                constexpr int e = decltype(e_)::value;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j % RING][e], breg[4 * j + e], acc, 0, 0, 0);
            });
            if constexpr (j + RING < NG) {
                __builtin_amdgcn_sched_barrier(0);
                a[j % RING] = bload16_sc1(rsA, voff, (j + RING) * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        return acc;
    };

#ifdef S2VT_AC_STAMP
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_readcyclecounter();
#define AC_STAMP(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); st_acc[i] += n_ - st_prev; st_prev = n_; } while (0)
#else
#define AC_STAMP(i) do { } while (0)
#endif
    for (int t = 0; t < T; ++t) {
        f32x4 acc = {ci[0], ci[1], ci[2], ci[3]};
        asm volatile("" : "+v"(acc));
        AC_STAMP(0);                                           // carried partial in registers
        // the NEXT step's carried partial is requested now (it is not touched by this step: the gates of step t go to slot t) and
        // lands under the step's hand-offs -- waited for at the top of a step it cost 1.5 us
        if (t + 1 < T) load_cinit(t + 1, cn);
        if (t > 0) {
            gs.wait_all((unsigned)(t - 1), pwave, lane);       // h_{t-1} and out_{t-1} of every workgroup are in the images
            AC_STAMP(1);                                       // end-of-step hand-off wait
            const float* hcur = g.himg + (size_t)(t & 1) * img_floats + (size_t)wave * NG * 256;
            if (roleQ) {
                // ---- query projection of this step: out_{t-1} @ Wa[:, tile], published first -- the attention role waits for it
                const float* qcur = g.qimg + (size_t)(t & 1) * img_floats + (size_t)wave * NG * 256;
                f32x4 hq = {0.f, 0.f, 0.f, 0.f};
                hq = pass_lds(qcur, Wq, hq);
                const int col = (int)blockIdx.x * 16 + l15;
                if (tok && col < H) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = wave * 16 + lq * 4 + r;
                        if (m < M)
                            __hip_atomic_store((gu32*)(g.hWa + (size_t)t * g.hwa_tstride + (size_t)m * H + col), __float_as_uint(hq[r]), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                gq.arrive_as((int)blockIdx.x, tid);
                AC_STAMP(2);                                   // query projection + publish
            }
            // ---- the recurrent block of the chain (needs nothing of this step)
            acc = pass_lds(hcur, Wl, acc);
            asm volatile("" : "+v"(acc));
            AC_STAMP(3);                                       // h block
        }
        if (roleA) {
            // ---- score / softmax / context of batch row brow (the arithmetic of attn.hip::attn_fwd_kernel)
            const int q = tid;                                  // 16-byte column group (H <= 1024: one per thread)
            const bool qok = 4 * q < H;
            const size_t rowoff = (size_t)brow * H + 4 * q;
            f32x4 pv[kACRows], vv[kACRows];
#pragma unroll
            for (int j = 0; j < kACRows; ++j) {                 // P and V of the first frames: issued before the wait (they do not depend on the step)
                pv[j] = f32x4{0.f, 0.f, 0.f, 0.f}; vv[j] = pv[j];
                if (j < Tv && qok) {
                    pv[j] = *reinterpret_cast<const f32x4*>(g.P + (size_t)j * M * H + rowoff);
                    vv[j] = *reinterpret_cast<const f32x4*>(g.Vt + (size_t)j * M * H + rowoff);
                }
            }
            f32x4 hv = {0.f, 0.f, 0.f, 0.f};
            if (t > 0) {
                gq.wait_all((unsigned)(t - 1), pwave, lane);
                const __amdgpu_buffer_rsrc_t rsH =
                    __builtin_amdgcn_make_buffer_rsrc(g.hWa + (size_t)t * g.hwa_tstride + (size_t)brow * H, 0, H * 4, 0x00020000);
                hv = bload16_sc1(rsH, q * 16, 0);               // (columns >= H: out of range, zeros)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                AC_STAMP(4);                                   // wait for hWa + its load (and P / V)
            }
            for (int c0 = 0; c0 < Tv; c0 += RCA) {
                const int nr = (Tv - c0) < RCA ? (Tv - c0) : RCA;
                for (int g0 = 0; g0 < nr; g0 += kACRows) {       // tanh rows of the chunk, kACRows frames' loads in flight at a time
                    if (c0 + g0 > 0) {
#pragma unroll
                        for (int j = 0; j < kACRows; ++j)
                            if (g0 + j < nr && qok) pv[j] = *reinterpret_cast<const f32x4*>(g.P + (size_t)(c0 + g0 + j) * M * H + rowoff);
                    }
                    if (qok) {
#pragma unroll
                        for (int j = 0; j < kACRows; ++j)
                            if (g0 + j < nr) {
                                f32x4 tt;
                                tt[0] = dm_tanhf(hv[0] + pv[j][0]); tt[1] = dm_tanhf(hv[1] + pv[j][1]);
                                tt[2] = dm_tanhf(hv[2] + pv[j][2]); tt[3] = dm_tanhf(hv[3] + pv[j][3]);
                                *reinterpret_cast<f32x4*>(Tt + (size_t)(g0 + j) * LDT + 4 * q) = tt;
                            }
                    }
                }
                __syncthreads();
                const int r = (tid & 63) * 4 + (tid >> 6);      // frame r of the chunk -> wave r & 3, lane r >> 2
                if (r < nr) ev[c0 + r] = score_chain(Tt + (size_t)r * LDT, wl, NG);      // (rows zero-padded to NG * 16: attn_score.h)
                __syncthreads();
            }
            AC_STAMP(5);                                       // tanh rows + score chains
            if (tid < Tv) xv[tid] = dm_expf(ev[tid]);
            __syncthreads();
            if (tid == 0) {
                float den = 0.f;
                for (int f = 0; f < Tv; ++f) den = den + xv[f];
                if (den == 0.f) den = den + 1.0f;
                sc[0] = den;
            }
            __syncthreads();
            if (tid < Tv) {
                const float al = xv[tid] / sc[0];
                g.alpha[((size_t)t * Tv + tid) * M + brow] = al;
                ev[tid] = al;
            }
            __syncthreads();
            if (tid == 0) {
                float s = 0.f;
                const int n8 = Tv < 8 ? Tv : 8;
                for (int f = 0; f < n8; ++f) s = s + ev[f];
                g.asum[(size_t)t * M + brow] = s;
            }
            if (qok) {
                f32x4 c4 = {0.f, 0.f, 0.f, 0.f};
                for (int c0 = 0; c0 < Tv; c0 += kACRows) {
                    const int nr = (Tv - c0) < kACRows ? (Tv - c0) : kACRows;
                    if (c0 > 0) {
#pragma unroll
                        for (int j = 0; j < kACRows; ++j)
                            if (j < nr) vv[j] = *reinterpret_cast<const f32x4*>(g.Vt + (size_t)(c0 + j) * M * H + rowoff);
                    }
#pragma unroll
                    for (int j = 0; j < kACRows; ++j)
                        if (j < nr) {
                            const float al = ev[c0 + j];
                            c4[0] = __builtin_fmaf(al, vv[j][0], c4[0]); c4[1] = __builtin_fmaf(al, vv[j][1], c4[1]);
                            c4[2] = __builtin_fmaf(al, vv[j][2], c4[2]); c4[3] = __builtin_fmaf(al, vv[j][3], c4[3]);
                        }
                }
                *reinterpret_cast<f32x4*>(Tt + 4 * q) = c4;              // staged (the tanh rows are free): regrouped into fragment slots below
                *reinterpret_cast<f32x4*>(g.ctx + (size_t)t * M * H + rowoff) = c4;      // the row-major history
            }
            {
                // ctx_t -> the fragment-order image as whole 16-byte slots: thread (group j = tid / 4, kq = tid % 4) takes k = 16 j + 4 e + kq
                __syncthreads();
                const int j = tid >> 2, kq = tid & 3;
                if (16 * j < H) {
                    u32x4v wv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) wv[e] = __float_as_uint(16 * j + 4 * e + kq < H ? Tt[16 * j + 4 * e + kq] : 0.0f);
                    const __amdgpu_buffer_rsrc_t rsCi = __builtin_amdgcn_make_buffer_rsrc(g.cimg, 0, 4 * NG * 1024, 0x00020000);
                    bstore16_sc1(rsCi, wv, (int)((((size_t)((brow >> 4) * NG + j) * 64 + (size_t)(kq * 16 + (brow & 15))) * 4) * 4), 0);
                }
            }
            gc.arrive_as(brow, tid);
            AC_STAMP(6);                                       // softmax + context + publish
        }
        gc.wait_all((unsigned)t, pwave, lane);
        AC_STAMP(7);                                           // wait for the context
        // ---- the context block of the chain, then BasicLSTMCell pointwise (gate order i, j, f, o; forget_bias 1.0): EPI_LSTM's expressions
        acc = pass_reg(g.cimg + (size_t)wave * NG * 256, acc);
        asm volatile("" : "+v"(acc));
        AC_STAMP(8);                                           // context block
#pragma unroll
        for (int r = 0; r < 4; ++r) zb[(lq * 4 + r) * ZS + l15] = acc[r];
        __builtin_amdgcn_wave_barrier();
        const f32x4 z = *reinterpret_cast<const f32x4*>(zb + rt * ZS + uu * 4);
        __builtin_amdgcn_wave_barrier();
        const float zi = z[0] + bi, zj = z[1] + bj, zf = z[2] + bf, zo = z[3] + bo;
        const float si = dm_sigmoidf(zi);
        const float tj = dm_tanhf(zj);
        const float sf = dm_sigmoidf(zf + 1.0f);
        const float so = dm_sigmoidf(zo);
        const float t1 = c_reg * sf;
        const float t2 = si * tj;
        const float cc = t1 + t2;
        const float hval = dm_tanhf(cc) * so;
        c_reg = cc;
        float oval = hval;
        if (drops && rok) oval = (hval / g.keep) * dropout_keep01(g.seed_lo, g.seed_hi, vid, sid, g.drop_code0 + (uint32_t)t, (uint32_t)u, g.keep);
        if (t + 1 < T) {
            if (rok) {
                __hip_atomic_store((gu32*)(g.himg + (size_t)((t + 1) & 1) * img_floats + a_own), __float_as_uint(hval), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (drops)
                    __hip_atomic_store((gu32*)(g.qimg + (size_t)((t + 1) & 1) * img_floats + a_own), __float_as_uint(oval), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            gs.arrive(tid);
        }
        AC_STAMP(9);                                           // pointwise + image stores + arrive
        if (rok) {
            const size_t o = (size_t)row * H + u;
            g.C[(size_t)(t + 1) * g.state_tstride + o] = c_reg;
            g.Hh[(size_t)(t + 1) * g.state_tstride + o] = hval;
            if (g.Out) g.Out[(size_t)(t + 1) * g.state_tstride + o] = oval;
            if (g.gates) {
                float* gp = g.gates + (size_t)t * g.gates_tstride + (size_t)row * 4 * H + u;
                gp[0] = si; gp[H] = tj; gp[2 * H] = sf; gp[3 * H] = so;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) ci[r] = cn[r];
        AC_STAMP(10);                                          // history stores issued, next partial requested
    }
#ifdef S2VT_AC_STAMP
    if (tid == 0 && (blockIdx.x == 0 || (int)blockIdx.x == nwg / 2 || (int)blockIdx.x == nwg - 1)) {
        const int slot = blockIdx.x == 0 ? 0 : ((int)blockIdx.x == nwg - 1 ? 2 : 1);
#pragma unroll
        for (int i = 0; i < 16; ++i) atomicAdd(&ac_stamp_acc[slot * 16 + i], st_acc[i]);
    }
#endif
}

struct ACfg { int ng; void (*fn)(const AttnChainKArgs); const char* name; };
const ACfg kACfg[] = {{8, attn_chain_kernel<8>, "attn_chain(ng8)"}, {64, attn_chain_kernel<64>, "attn_chain(ng64)"}};
constexpr int kNumACfg = 2;

int ac_lds_bytes(int ng) { return (2 * ng * 256 + 4 * 16 * 20 + kACRows * (ng * 16 + 4) + ng * 16 + 2 * kACMaxTv + 4) * 4; }

struct ADev {
    std::once_flag once;
    bool ok = false;
    int per_cu[kNumACfg] = {};
};
constexpr int kMaxDev = 32;
ADev g_adev[kMaxDev];
ADev* adev_state()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
    ADev& d = g_adev[dev];
    std::call_once(d.once, [&d] {
        bool ok = true;
        for (int i = 0; ok && i < kNumACfg; ++i) {
            const int lds = ac_lds_bytes(kACfg[i].ng);
            ok = hipFuncSetAttribute(reinterpret_cast<const void*>(kACfg[i].fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
            int n = 0;
            if (ok && hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kACfg[i].fn), 256, lds) == hipSuccess) d.per_cu[i] = n;
        }
        d.ok = ok;
    });
    return &d;
}

int ac_cfg(int H) { return (H + 15) / 16 <= 8 ? 0 : 1; }

}  // namespace

bool attn_chain_eligible(int B, int H, int Tv)
{
    static const bool off = [] { const char* e = getenv("S2VT_ACHAIN"); return e && e[0] == '0'; }();      // dev / test knob: per-step launches
    if (off || chain_persistent_disabled()) return false;
    ChainHost hst;
    ADev* d = adev_state();
    if (!chain_host(&hst) || !d || !d->ok) return false;
    if (!(B >= 1 && B <= 64 && H >= 16 && (H & 3) == 0 && H <= 1024 && Tv >= 1 && Tv <= kACMaxTv)) return false;
    const int grid = H / 4;
    if (grid > hst.num_cus || B > grid) return false;          // one workgroup per CU; a batch row per attention workgroup
    return (long)d->per_cu[ac_cfg(H)] * hst.num_cus >= grid;    // every workgroup resident at once
}

size_t attn_chain_scratch_floats(int H)
{
    const int ng = kACfg[ac_cfg(H)].ng;
    return (size_t)5 * 4 * ng * 256;                            // h and dropped-out images (two parities each) + the context image
}

hipError_t launch_attn_chain(const AttnChainLaunch& a, hipStream_t st)
{
    if (!attn_chain_eligible(a.B, a.H, a.Tv)) return hipErrorInvalidValue;
    if (a.T <= 0) return hipSuccess;
    const auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    if (!al16(a.W3) || (a.ldw & 3) || !al16(a.Wa) || (a.ldwa & 3) || !al16(a.P) || !al16(a.Vt) || !al16(a.hWa) || !al16(a.ctx) || !al16(a.img) ||
        (a.hwa_tstride & 3))
        return hipErrorInvalidValue;
    ChainHost hst;
    if (!chain_host(&hst)) return hipErrorInvalidValue;
    const int ci = ac_cfg(a.H), ng = kACfg[ci].ng;
    AttnChainKArgs k;
    std::memset(&k, 0, sizeof(k));
    k.W3 = a.W3; k.ldw = a.ldw; k.b3 = a.b3; k.cinit = a.cinit; k.cinit_tstride = a.cinit_tstride; k.ldcinit = a.ldcinit;
    k.C = a.C; k.Hh = a.Hh; k.Out = a.Out; k.state_tstride = a.state_tstride; k.gates = a.gates; k.gates_tstride = a.gates_tstride;
    k.Wa = a.Wa; k.ldwa = a.ldwa; k.P = a.P; k.Vt = a.Vt; k.w = a.w; k.hWa = a.hWa; k.hwa_tstride = a.hwa_tstride;
    k.alpha = a.alpha; k.asum = a.asum; k.ctx = a.ctx; k.B = a.B; k.H = a.H; k.T = a.T; k.Tv = a.Tv;
    k.keep = a.keep; k.seed_lo = a.seed_lo; k.seed_hi = a.seed_hi; k.drop_code0 = a.drop_code0; k.video_id = a.video_id; k.sample_id = a.sample_id;
    const size_t imgf = (size_t)4 * ng * 256;
    k.himg = a.img; k.qimg = a.keep < 1.0f ? a.img + 2 * imgf : a.img; k.cimg = a.img + 4 * imgf;
    k.sync = a.sync; k.status = hst.status_dev; k.fault = hst.fault; k.spin_limit = hst.spin_limit;
    ChainLaunchOrder order;                                    // one persistent grid at a time per process
    {
        hipError_t we = order.before(st, hst.device);
        if (we != hipSuccess) return we;
    }
    ZeroList z;
    z.add(a.sync, kAttnChainSyncBytes); z.add(a.img, 5 * imgf * 4);       // (rows >= B and k >= H of the images must read as zeros)
    hipError_t e = launch_zero_regions(z, st);
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)(a.H / 4));
    const double flops = (2.0 * a.B * (double)a.H * 4.0 * a.H * 2.0 + 2.0 * a.B * (double)a.H * a.H) * a.T;     // h and context blocks of LSTM3 + the query projection
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool prof = prof_wants(9, ci);
    if (prof) {
        hipError_t pe = prof_events(&e0, &e1);
        if (pe != hipSuccess) return pe;
        (void)hipEventRecord(e0, st);
    }
    hipLaunchKernelGGL(kACfg[ci].fn, grid, dim3(256), ac_lds_bytes(ng), st, k);
    if (prof) {
        (void)hipEventRecord(e1, st);
        prof_record(9, ci, kACfg[ci].name, flops, e0, e1);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return order.after(st, hst.device);
}

}  // namespace s2vt

#ifdef S2VT_AC_STAMP
// dev build only: per-phase clock sums of attn_chain_kernel (3 workgroups x 16 phases), read and reset (tools/ac_stamp.py)
extern "C" int s2vt_ac_stamp_read(unsigned long long* out48)
{
    if (hipDeviceSynchronize() != hipSuccess) return -4;
    if (hipMemcpyFromSymbol(out48, HIP_SYMBOL(s2vt::ac_stamp_acc), 48 * sizeof(unsigned long long)) != hipSuccess) return -4;
    unsigned long long z[48] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(s2vt::ac_stamp_acc), z, sizeof(z)) == hipSuccess ? 0 : -4;
}
#endif
