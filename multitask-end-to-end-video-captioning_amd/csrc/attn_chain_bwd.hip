// attn_chain_bwd.hip -- back-propagation through the temporal-attention captioner's recurrence (tf.gradients through
// original_attention.py:109-135) in ONE persistent launch.  Iteration t = T-1 .. 0:
//     (P) dout_t = d(output layer)[t] + dq_{t+1};  dh_t = dropout'(dout_t) + dh_rec;  BasicLSTMCell backward -> dz_t, dc
//     (M) [dh_rec | dctx_t] = dz_t @ [W3[2H:3H] ; W3[0:H]]^T            (K = 4H: one gate's quarter per workgroup, 4-workgroup exchange)
//     (A) attention backward of step t on dctx_t + d(output layer)'s context block -> dhWa_t, dP +=, dV +=, dw +=  (one batch row per workgroup)
//     (Q) dq_t = dhWa_t @ Wa^T                                         (gradient w.r.t. out_{t-1} through step t's query)
// As launches this was four kernels and ~70 us per step (a pointwise launch, two skinny split-K products, the attention
// backward).  Here it is chain_bwd.hip's construction with the attention phases added:
//   * workgroup (j, g) owns 16 hidden units and gate g's quarter of the reduction of BOTH blocks of (M): its [H x 16] slices
//     W3[2H + 16j .., gH .. gH+H) and W3[16j .., gH .. gH+H) sit in LDS (2 x 64 KB) in B-fragment order; dz_t crosses the chip as
//     four per-gate images in A-fragment order; the four gate partials of a unit group meet through an exchange among the four
//     workgroups (j, 0..3); workgroup (j, g) then owns row tile g of its 16 units: dh_rec and dq stay in a register of the thread
//     that finishes (row, unit), dc for all T steps;
//   * the summed dctx_t goes out row-major; the B attention workgroups (one batch row each) run the arithmetic of
//     attn.hip::attn_bwd_kernel and publish dhWa_t row-major (the history the batched dWa contraction reads) and as a
//     fragment-order image;
//   * (Q) is one 16 x 16 tile per workgroup -- its own row tile and units: the K = H reduction is cut over the four waves (order-
//     free), each wave's quarter of Wa[16j .., :] in 4 NG / 4 registers per lane;
//   * hand-offs per iteration, three in series: dz images (all, sharded counter); the partial tiles of (M) -- ONE arrival for the 4-workgroup
//     exchange (cluster counter: dh_rec) and for the attention role (a second sharded counter), which sums the four gate partials of the
//     context block itself; dhWa (attention workgroups); the sc1 form of chain_common.h.
// The embedding block of dz @ W3^T does not feed the recurrence: the caller computes it for all steps at once afterwards.
// Gradients are order-free fp32 (checked against float64 autograd, tests/test_gpu_attention_model.py).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "chain_common.h"

namespace s2vt {

namespace {

constexpr int kABMaxTv = 64;
constexpr int kABRegFrames = 5;                            // up to this many frames a row's P, V, dP, dV live in registers (the script's default Tv)

struct AttnBwdChainKArgs {
    const float* W3; int ldw;                          // [3H, 4H]
    const float* Wa; int ldwa;                         // [H, H]
    const float* gates; size_t gates_tstride;          // activated gates [T][B][4H]
    const float* C; size_t state_tstride;              // cell states [T+1][B][H]
    const float* dcat; size_t dcat_tstride; int ld_cat;   // d[out | ctx | emb] of the output layer [T][B][3H]
    float* dctx_hist;                                  // Tv > 5: == dcat, whose ctx block of step t is OVERWRITTEN with the total d(ctx_t) (attn_dpdv_kernel reads it)
    float* deh;                                        // Tv > 5: [T][Tv][B] d(score) of every step (the same)
    float* dZ; size_t dz_tstride;                      // [T][B][4H]
    const float* hWa; size_t hwa_tstride;              // forward history [T][B][H] (slot 0 unused: zero query)
    const float* P; const float* Vt; const float* w; const float* alpha;   // [Tv,B,H] x2, [H], [T][Tv][B]
    const float* reg_coef; const float* asum; float reg_m;                 // alpha regulariser: [T*B] or NULL, [T][B], m
    float* dhWa; size_t dhwa_tstride;                  // history [T][B][H], slots >= 1 written
    float* dP; float* dVt; float* dw;                  // accumulated [Tv,B,H] x2, [H]
    int B, H, T, Tv;
    float keep; uint32_t seed_lo, seed_hi, drop_code0;
    const int32_t* video_id; const int32_t* sample_id;
    float* img;                                        // 2 parities x 4 gate images of dz, [4 row tiles][NG][256] each
    float* ex;                                         // [unit groups][4 gates][8 tiles][256]
    float* dctxs;                                      // [64][H] the recurrence's part of d(ctx_t), row-major
    float* qimg;                                       // dhWa_t in A-fragment order [4][NG][256]
    unsigned* sync;                                    // dz | dctx | dhWa counter sets (kChainSyncBytes each) | one line per unit group
    unsigned* status; unsigned* fault; unsigned spin_limit;
    int ncg;
};

#ifdef S2VT_AC_STAMP
__device__ unsigned long long ab_stamp_acc[3 * 16];          // dev build: per-phase clock sums of three workgroups
#endif
template <int NG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_bwd_chain_kernel(const AttnBwdChainKArgs g)
{
    constexpr int ZS = 20;
    constexpr int RING = NG < 16 ? NG : 16;
    constexpr int QG = NG / 4;                                 // k-groups of the query product per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                                          // [NG][2 blocks][64][4]: B fragments (block 0: h rows, block 1: context rows)
    const int tid = threadIdx.x, lane = tid & 63;
    const int pwave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* zb = smem + NG * 2 * 256 + pwave * (16 * ZS);       // per-wave transpose tile
    float* dzl = smem + NG * 2 * 256 + 4 * 16 * ZS;            // [4 gates][16 rows][17]
    float* cq = dzl + 4 * 16 * 17;                             // [4 waves][256] partial tiles of the query product
    float* dcl = cq + 4 * 256;                                 // [NG * 16] d(ctx) of the attention role's row
    float* dal = dcl + NG * 16;                                // [64]
    float* del = dal + kABMaxTv;                               // [64]
    float* all_ = del + kABMaxTv;                              // [64]
    const int l15 = lane & 15, lq = lane >> 4;
    const int H = g.H, M = g.B, T = g.T, Tv = g.Tv;
    const int wg = (int)blockIdx.x;
    const int gate = (wg >> 3) & 3;
    const int jj = (wg >> 5) * 8 + (wg & 7);
    if (jj >= g.ncg) return;                                   // (grid padded to whole groups of 8)
    const int u0 = jj * 16;
    const size_t img_floats = (size_t)4 * NG * 256;            // one gate image
    const int lin = jj * 4 + gate;
    const int brow = 4 * g.ncg - 1 - lin;                      // attention role: batch row
    const bool roleA = brow < M;
    // Every workgroup of a row tile reads the SAME dhWa image at the same moment: walked in the same k-group order by all of them, the 63
    // readers of a 1-KB block queue on the few L2 channels that hold it (s_memtime stamps, round 5: the 16 fragment loads of the query product
    // took 10 000 cycles; 3 500 with the rotation).  The product is order-free (a gradient), so every unit group starts its walk at a different
    // k-group: register slot j of the wave's quarter of Wa holds group (j + rotq) mod QG.  (The same rotation on the dz images of phase (M)
    // measured SLOWER, 19 300 -> 22 600 cycles for the 512 MFMAs: there the readers of a block are spread over the ring's 16 loads in flight
    // anyway, and walking together is what lets 31 of an XCD's 32 readers hit the line the first one fetched.)
    const int rotq = jj & (QG - 1);

    // ---- this workgroup's two slices -> LDS, once.  B[k][n] = W3[base + u0 + n][gate * H + k]; fragment slot (group, block, lane L = i * 16 + n)
    // holds k = 16 group + 4 e + i for e = 0 .. 3.  Destination-indexed: a thread fills whole 16-byte slots, consecutive threads consecutive slots
    // (source-indexed, 4-byte writes of one loaded row vector landed 64 lanes on 4 banks: the bulk of this kernel's LDS bank conflicts)
    for (int sl = tid; sl < NG * 2 * 64; sl += 256) {
        const int L = sl & 63, blk = (sl >> 6) & 1, grp = sl >> 7;
        const int i = L >> 4, n = L & 15;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (u0 + n < H) {
            const float* row = g.W3 + (size_t)((blk == 0 ? 2 * H : 0) + u0 + n) * g.ldw + (size_t)gate * H + 16 * grp + i;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (16 * grp + 4 * e + i < H) v[e] = row[4 * e];
        }
        *reinterpret_cast<f32x4*>(Wl + (size_t)sl * 4) = v;
    }
    // this wave's quarter of Wa[u0 + l15][:] -> registers: k-step s of the quarter holds Wa[u0 + l15][16 (QG w + s / 4) + 4 (s % 4) + lq]
    float wa[4 * QG];
#pragma unroll
    for (int s = 0; s < 4 * QG; ++s) {
        const int k = 16 * (QG * pwave + (((s >> 2) + rotq) & (QG - 1))) + 4 * (s & 3) + lq;
        wa[s] = (u0 + l15 < H && k < H) ? g.Wa[(size_t)(u0 + l15) * g.ldwa + k] : 0.0f;
    }

    // ---- the (row, unit) this thread finishes at every step: row tile `gate`, 16 units
    const int pr = tid >> 4, pn = tid & 15;
    const int pm = gate * 16 + pr, pu = u0 + pn;
    const bool pok = pm < M && pu < H;
    float dc_reg = 0.0f;
    float cnew = pok ? g.C[(size_t)T * g.state_tstride + (size_t)pm * H + pu] : 0.0f;       // c_{T-1}
    const uint32_t vid = (g.keep < 1.0f && pok) ? (uint32_t)g.video_id[pm] : 0u;
    const uint32_t sid = (g.keep < 1.0f && pok) ? (uint32_t)g.sample_id[pm] : 0u;
    gu32* const base = (gu32*)g.sync;
    gu32* const ccount = base + 3 * (kChainSyncBytes / 4) + jj * 32;      // the unit group's exchange counter
    const __amdgpu_buffer_rsrc_t rsEx = __builtin_amdgcn_make_buffer_rsrc(g.ex, 0, g.ncg * 4 * 8 * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsImg = __builtin_amdgcn_make_buffer_rsrc(g.img, 0, (int)(8 * img_floats * 4), 0x00020000);
    GridSync gs{base, g.status, g.fault, g.spin_limit, g.ncg, false, 4u};                       // dz images
    GridSync gd{base + kChainSyncBytes / 4, g.status, g.fault, g.spin_limit, g.ncg, false, 4u};   // summed dctx
    GridSync gh{base + 2 * (kChainSyncBytes / 4), g.status, g.fault, g.spin_limit, M, false};      // dhWa: the attention workgroups arrive (sharded:
                                                                                                   // one word polled by every workgroup starves its own arrivals)
    const bool wok = pwave * 16 < M;                           // MFMA side: this wave's row tile holds rows of the problem
    const int voff = wok ? lane * 16 : (int)0x80000000u;
    __syncthreads();

    float sg[4], cprev, dxo;
    auto load_step = [&](int t) __attribute__((always_inline)) {     // operands of step t's pointwise part (independent of the recurrence)
        const float* gp = g.gates + (size_t)t * g.gates_tstride + (size_t)pm * 4 * H + pu;
#pragma unroll
        for (int q = 0; q < 4; ++q) sg[q] = pok ? gp[(size_t)q * H] : 0.0f;
        cprev = pok ? g.C[(size_t)t * g.state_tstride + (size_t)pm * H + pu] : 0.0f;
        dxo = pok ? g.dcat[(size_t)t * g.dcat_tstride + (size_t)pm * g.ld_cat + pu] : 0.0f;
    };
    load_step(T - 1);

    // attention role with few frames (the script's default 5): the row's P and V, and the accumulators of dP and dV, live in REGISTERS
    // for the whole launch (4 columns x Tv frames each per thread) -- as global read-modify-writes per iteration they were 7 us of it
    constexpr int TVR = kABRegFrames;
    const bool aregs = roleA && Tv <= TVR;
    const bool aqok = 4 * tid < H;
    f32x4 pR[TVR], vR[TVR], dPR[TVR], dVR[TVR];
#pragma unroll
    for (int f = 0; f < TVR; ++f) {
        pR[f] = vR[f] = dPR[f] = dVR[f] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (aregs && aqok && f < Tv) {
            pR[f] = *reinterpret_cast<const f32x4*>(g.P + ((size_t)f * M + brow) * H + 4 * tid);
            vR[f] = *reinterpret_cast<const f32x4*>(g.Vt + ((size_t)f * M + brow) * H + 4 * tid);
        }
    }
    f32x4 dwR = {0.f, 0.f, 0.f, 0.f};                          // d(score vector) of this thread's four columns, summed over the iterations (one
                                                               // atomicAdd per column at the END: 64 workgroups x 1000 contended atomics per iteration cost ~5 us of it)
    float dh_rec = 0.0f, dq = 0.0f;
    unsigned it = 0;                                           // iterations done
#ifdef S2VT_AC_STAMP
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_readcyclecounter();
#define AB_STAMP(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); st_acc[i] += n_ - st_prev; st_prev = n_; } while (0)
#else
#define AB_STAMP(i) do { } while (0)
#endif
    for (int t = T - 1; t >= 0; --t, ++it) {
        // ---- (P) BasicLSTMCell backward pointwise (the expressions of lstm_bwd_pointwise_kernel)
        float dzv[4];
        {
            float d = dxo + dq;
            if (g.keep < 1.0f) d = (d / g.keep) * dropout_keep01(g.seed_lo, g.seed_hi, vid, sid, g.drop_code0 + (uint32_t)t, (uint32_t)pu, g.keep);
            const float dht = dh_rec + d;
            const float si = sg[0], tj = sg[1], sf = sg[2], so = sg[3];
            const float tc = dm_tanhf(cnew);
            const float dc = dht * so * (1.f - tc * tc) + dc_reg;
            dzv[0] = dc * tj * si * (1.f - si);
            dzv[1] = dc * si * (1.f - tj * tj);
            dzv[2] = dc * cprev * sf * (1.f - sf);
            dzv[3] = dht * tc * so * (1.f - so);
            dc_reg = dc * sf;
            cnew = cprev;
        }
        // dz_t -> the four gate images (regrouped through LDS: every thread writes ONE 16-byte fragment slot), then the history
        {
            const size_t inext = (size_t)(t & 1) * 4 * img_floats;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) dzl[(q * 16 + pr) * 17 + pn] = pok ? dzv[q] : 0.0f;
            __syncthreads();
            const int q = tid >> 6, L = tid & 63, r = L & 15, kq = L >> 4;
            u32x4v wv;
#pragma unroll
            for (int e = 0; e < 4; ++e) wv[e] = __float_as_uint(dzl[(q * 16 + r) * 17 + kq + 4 * e]);
            const size_t dst = inext + (size_t)q * img_floats + ((size_t)(gate * NG + jj) * 64 + L) * 4;
            bstore16_sc1(rsImg, wv, (int)(dst * 4), 0);
            gs.arrive(tid);
        }
        AB_STAMP(0);                                           // pointwise + dz images + arrive
        if (pok) {
            float* zp = g.dZ + (size_t)t * g.dz_tstride + (size_t)pm * 4 * H + pu;
#pragma unroll
            for (int q = 0; q < 4; ++q) zp[(size_t)q * H] = dzv[q];
        }
        if (t > 0) load_step(t - 1);
        AB_STAMP(1);                                           // history, next operands requested

        // ---- (M) dz_t[:, gate block] @ [h rows ; context rows]^T for this wave's row tile
        gs.wait_all(it, pwave, lane);
        AB_STAMP(2);                                           // dz hand-off wait
        {
            const float* acur = g.img + (size_t)(t & 1) * 4 * img_floats + (size_t)gate * img_floats + (size_t)pwave * NG * 256;
            const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(acur), 0, NG * 1024, 0x00020000);
            f32x4 a[RING];
            f32x4 acc[2][2];
            acc[0][0] = acc[0][1] = acc[1][0] = acc[1][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            static_for<0, RING>([&](auto j_) { constexpr int j = decltype(j_)::value; a[j] = bload16_sc1(rsA, voff, j * 1024); });
            __builtin_amdgcn_sched_barrier(0);
            const f32x4* bl = reinterpret_cast<const f32x4*>(Wl) + lane;
            constexpr int PB = NG < 4 ? NG : 4;
            f32x4 b[PB][2];
            static_for<0, PB>([&](auto j_) { constexpr int j = decltype(j_)::value; b[j][0] = bl[(j * 2) * 64]; b[j][1] = bl[(j * 2 + 1) * 64]; });
            static_for<0, NG>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                const f32x4 b0 = b[j % PB][0], b1 = b[j % PB][1];
                if constexpr (j + PB < NG) { b[j % PB][0] = bl[((j + PB) * 2) * 64]; b[j % PB][1] = bl[((j + PB) * 2 + 1) * 64]; }
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, 4>([&](auto e_) {
                    constexpr int e = decltype(e_)::value;
                    acc[e & 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j % RING][e], b0[e], acc[e & 1][0], 0, 0, 0);
                    acc[e & 1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j % RING][e], b1[e], acc[e & 1][1], 0, 0, 0);
                });
                if constexpr (j + RING < NG) {
                    __builtin_amdgcn_sched_barrier(0);
                    a[j % RING] = bload16_sc1(rsA, voff, (j + RING) * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
            AB_STAMP(3);                                       // MFMAs
            // partial tiles -> the unit group's exchange [gate][row tile][block] (row-major 16 x 16, one 16-byte store per lane)
            const size_t exc = (size_t)jj * 4 * 8;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int r = 0; r < 4; ++r) zb[(lq * 4 + r) * ZS + l15] = acc[0][c][r] + acc[1][c][r];
                __builtin_amdgcn_wave_barrier();
                const f32x4 row = *reinterpret_cast<const f32x4*>(zb + (lane >> 2) * ZS + (lane & 3) * 4);
                __builtin_amdgcn_wave_barrier();
                bstore16_sc1(rsEx, __builtin_bit_cast(u32x4v, row), (int)(((exc + (size_t)gate * 8 + (size_t)pwave * 2 + c) * 256 + lane * 4) * 4), 0);
            }
            // ONE arrival serves both consumers of the partial tiles (round 5: three hand-offs per iteration in series instead of four): the
            // unit group's exchange (dh_rec, below) and the attention role, which used to wait for the exchanged SUM of the context block
            // -- exchange, sum, store, drain, a second arrival -- and now sums the four gate partials itself, in the same gate order
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(gd.sync + (blockIdx.x & (kShards - 1)) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(ccount, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            AB_STAMP(5);                                       // partial tiles published (exchange + dctx arrival)
            gs.wait_one(ccount, 4u * (it + 1u), pwave, lane);
            float s0 = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                s0 += __uint_as_float(__hip_atomic_load((const gu32*)(g.ex + (exc + (size_t)q * 8 + gate * 2) * 256 + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            AB_STAMP(4);                                       // 4-workgroup exchange
            dh_rec = s0;                                       // gradient w.r.t. h_{t-1} through step t's recurrent rows
        }
        // ---- (A) attention backward of step t for batch row brow (attn.hip::attn_bwd_kernel's arithmetic)
        if (roleA) {
            const int q4 = tid;                                 // 16-byte column group (H <= 1024)
            const bool qok = 4 * q4 < H;
            const size_t rowoff = (size_t)brow * H + 4 * q4;
            f32x4 dense = {0.f, 0.f, 0.f, 0.f}, hv = {0.f, 0.f, 0.f, 0.f}, wh = {0.f, 0.f, 0.f, 0.f};
            if (qok) {                                          // (nothing here depends on this iteration's recurrence: issued before the wait)
                dense = *reinterpret_cast<const f32x4*>(g.dcat + (size_t)t * g.dcat_tstride + (size_t)brow * g.ld_cat + H + 4 * q4);
                if (t > 0) hv = *reinterpret_cast<const f32x4*>(g.hWa + (size_t)t * g.hwa_tstride + rowoff);
                wh = *reinterpret_cast<const f32x4*>(g.w + 4 * q4);
            }
            if (tid < Tv) all_[tid] = g.alpha[((size_t)t * Tv + tid) * M + brow];
            // few frames: tanh(hWa_t + P_f) of this thread's columns does not depend on this iteration's recurrence either -- computed while the
            // other workgroups finish their products, not behind the wait (20 tanh per thread: ~3 000 cycles of the iteration's critical path)
            float tnR[TVR][4];
            if (aregs) {
#pragma unroll
                for (int f = 0; f < TVR; ++f)
#pragma unroll
                    for (int i = 0; i < 4; ++i) tnR[f][i] = (f < Tv && qok) ? dm_tanhf(hv[i] + pR[f][i]) : 0.0f;
            }
            gd.wait_all(it, pwave, lane);
            {
                // the recurrence's part of d(ctx_t) of this row: the context-block partials of the four gate workgroups of unit group
                // q4 / 4, row brow of row tile brow / 16, summed in gate order from +0 (the sum the exchange used to publish)
                const int jx = q4 >> 2;
                f32x4 pq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int off = qok ? (int)(((((size_t)jx * 4 + q) * 8 + (size_t)(brow >> 4) * 2 + 1) * 256 + (size_t)(brow & 15) * 16 + (size_t)(q4 & 3) * 4) * 4)
                                        : (int)0x80000000u;
                    pq[q] = bload16_sc1(rsEx, off, 0);
                }
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[0] += pq[q][0]; v[1] += pq[q][1]; v[2] += pq[q][2]; v[3] += pq[q][3]; }
                dense[0] += v[0]; dense[1] += v[1]; dense[2] += v[2]; dense[3] += v[3];
            }
            if (aregs) {
                // dalpha[f] = <dctx, V[f, b, :]>: every thread's four columns of every frame, then one reduction over the workgroup
                float part[TVR];
#pragma unroll
                for (int f = 0; f < TVR; ++f) {
                    float s = dense[0] * vR[f][0] + dense[1] * vR[f][1] + dense[2] * vR[f][2] + dense[3] * vR[f][3];
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                    part[f] = s;
                }
                __syncthreads();                                // (cq: free between two query products)
                if (lane == 0) {
#pragma unroll
                    for (int f = 0; f < TVR; ++f) cq[pwave * 8 + f] = part[f];
                }
                __syncthreads();
                AB_STAMP(6);                                   // dctx wait + load
                if (tid < Tv) dal[tid] = (cq[tid] + cq[8 + tid]) + (cq[16 + tid] + cq[24 + tid]);
            } else {
                if (qok) {
                    *reinterpret_cast<f32x4*>(dcl + 4 * q4) = dense;
                    // the total d(ctx_t) of this row: what the accumulation of dV needs, done for all steps at once behind the launch
                    *reinterpret_cast<f32x4*>(g.dctx_hist + (size_t)t * g.dcat_tstride + (size_t)brow * g.ld_cat + H + 4 * q4) = dense;
                }
                __syncthreads();
                AB_STAMP(6);                                   // dctx wait + load
                for (int f = pwave; f < Tv; f += 16) {          // dalpha[f] = <dctx, V[f, b, :]>: a wave takes frames f, f+4, f+8, f+12 together,
                    f32x4 xv[4][4];                             // and ALL their loads go out before the first dot (16 in flight per lane: as
                                                                // 4 x 4 dependent batches the rows' L2 latency was 12 us of a 52 us iteration at 32 frames)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int q = lane + 64 * k;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            xv[k][u] = f32x4{0.f, 0.f, 0.f, 0.f};
                            if (q < (H >> 2) && f + 4 * u < Tv)
                                xv[k][u] = *reinterpret_cast<const f32x4*>(g.Vt + ((size_t)(f + 4 * u) * M + brow) * H + 4 * q);
                        }
                    }
                    float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int q = lane + 64 * k;
                        if (q < (H >> 2)) {
                            const f32x4 d = *reinterpret_cast<const f32x4*>(dcl + 4 * q);
#pragma unroll
                            for (int u = 0; u < 4; ++u) s4[u] += d[0] * xv[k][u][0] + d[1] * xv[k][u][1] + d[2] * xv[k][u][2] + d[3] * xv[k][u][3];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        float s = s4[u];
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                        if (lane == 0 && f + 4 * u < Tv) dal[f + 4 * u] = s;
                    }
                }
            }
            __syncthreads();
            AB_STAMP(7);                                       // dalpha dots
            if (tid < 64) {                                      // softmax backward over the Tv <= 64 frames: one lane per frame (a serial loop
                float al = 0.f, dl = 0.f;                      // of one thread was 2.3 us of the iteration at 32 frames)
                if (tid < Tv) {
                    al = all_[tid]; dl = dal[tid];
                    if (g.reg_coef && tid < 8 && (g.reg_m - g.asum[(size_t)t * M + brow]) > 0.f) dl -= g.reg_coef[(size_t)t * M + brow];
                }
                float dot = al * dl;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
                if (tid < Tv) del[tid] = al * (dl - dot);
            }
            __syncthreads();
            if (!aregs && tid < Tv) g.deh[((size_t)t * Tv + tid) * M + brow] = del[tid];
            AB_STAMP(8);                                       // de
            if (qok) {
                f32x4 accq = {0.f, 0.f, 0.f, 0.f}, dwl = {0.f, 0.f, 0.f, 0.f};
                if (aregs) {
#pragma unroll
                    for (int f = 0; f < TVR; ++f)
                        if (f < Tv) {
                            const float d = del[f], alt = all_[f];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float Tn = tnR[f][i];
                                const float s_ = d * wh[i] * (1.f - Tn * Tn);
                                accq[i] += s_;
                                dwl[i] += d * Tn;
                                dPR[f][i] += s_;
                                dVR[f][i] += alt * dense[i];
                            }
                        }
                } else
                // more frames than registers hold: P is streamed (16 frames' loads in flight), and the accumulation of dP and dV -- as
                // global read-modify-writes here they were two thirds of this phase's 0.77 MB per iteration through one CU's L2 port
                // at 32 frames -- is left to attn_dpdv_kernel, which redoes the tanh for all steps at once on the whole chip
                for (int f0 = 0; f0 < Tv; f0 += 16) {
                    f32x4 pv[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        if (f0 + j < Tv) pv[j] = *reinterpret_cast<const f32x4*>(g.P + (size_t)(f0 + j) * M * H + rowoff);
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        if (f0 + j < Tv) {
                            const float d = del[f0 + j];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float Tn = dm_tanhf(hv[i] + pv[j][i]);
                                const float s_ = d * wh[i] * (1.f - Tn * Tn);
                                accq[i] += s_;
                                dwl[i] += d * Tn;
                            }
                        }
                }
                if (t > 0) {
                    *reinterpret_cast<f32x4*>(dcl + 4 * q4) = accq;                 // staged: regrouped into fragment slots below
                    *reinterpret_cast<f32x4*>(g.dhWa + (size_t)t * g.dhwa_tstride + rowoff) = accq;      // the row-major history
                }
                dwR[0] += dwl[0]; dwR[1] += dwl[1]; dwR[2] += dwl[2]; dwR[3] += dwl[3];
            }
            if (t > 0) {
                // dhWa_t -> the fragment-order image as whole 16-byte slots: thread (group j = tid / 4, kq = tid % 4) takes k = 16 j + 4 e + kq
                __syncthreads();
                const int j = tid >> 2, kq = tid & 3;
                if (16 * j < H) {
                    u32x4v wv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) wv[e] = __float_as_uint(16 * j + 4 * e + kq < H ? dcl[16 * j + 4 * e + kq] : 0.0f);
                    const __amdgpu_buffer_rsrc_t rsQi = __builtin_amdgcn_make_buffer_rsrc(g.qimg, 0, 4 * NG * 1024, 0x00020000);
                    bstore16_sc1(rsQi, wv, (int)((((size_t)((brow >> 4) * NG + j) * 64 + (size_t)(kq * 16 + (brow & 15))) * 4) * 4), 0);
                }
                gh.arrive_as(brow, tid);
            }
            AB_STAMP(9);                                       // main loop + dhWa publish
        }
        // ---- (Q) dq_t = dhWa_t @ Wa^T for this workgroup's own tile (row tile `gate`, its 16 units): K cut over the four waves
        if (t > 0) {
            gh.wait_all(it, pwave, lane);
            AB_STAMP(10);                                      // dhWa wait
            const float* qt = g.qimg + (size_t)gate * NG * 256 + (size_t)(QG * pwave) * 256;
            const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qt), 0, QG * 1024, 0x00020000);
            const int vq = gate * 16 < M ? lane * 16 : (int)0x80000000u;
            f32x4 aq[QG];
            static_for<0, QG>([&](auto j_) { constexpr int j = decltype(j_)::value; aq[j] = bload16_sc1(rsQ, vq, ((j + rotq) & (QG - 1)) * 1024); });
            // all 16 fragments first: left to its own waits hipcc interleaved them with the MFMAs in a way that cost 2.4 us here (stamps)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            AB_STAMP(12);                                      // query product: fragments landed
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
            static_for<0, QG>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[j][0], wa[4 * j + 0], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[j][1], wa[4 * j + 1], c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[j][2], wa[4 * j + 2], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[j][3], wa[4 * j + 3], c1, 0, 0, 0);
            });
#ifdef S2VT_AC_STAMP
            asm volatile("" : "+v"(c0), "+v"(c1));
            AB_STAMP(13);                                      // (dev) query product: MFMAs
#endif
#pragma unroll
            for (int r = 0; r < 4; ++r) cq[pwave * 256 + (lq * 4 + r) * 16 + l15] = c0[r] + c1[r];
            __syncthreads();
            dq = (cq[tid] + cq[256 + tid]) + (cq[512 + tid] + cq[768 + tid]);
            AB_STAMP(11);                                      // query product
        }
    }
    if (roleA && aqok) {
        atomicAdd(g.dw + 4 * tid, dwR[0]); atomicAdd(g.dw + 4 * tid + 1, dwR[1]); atomicAdd(g.dw + 4 * tid + 2, dwR[2]); atomicAdd(g.dw + 4 * tid + 3, dwR[3]);
    }
    if (aregs && aqok) {
#pragma unroll
        for (int f = 0; f < TVR; ++f)
            if (f < Tv) {
                const size_t o = ((size_t)f * M + brow) * H + 4 * tid;
                f32x4 a0 = *reinterpret_cast<const f32x4*>(g.dP + o), a1 = *reinterpret_cast<const f32x4*>(g.dVt + o);
#pragma unroll
                for (int i = 0; i < 4; ++i) { a0[i] += dPR[f][i]; a1[i] += dVR[f][i]; }
                *reinterpret_cast<f32x4*>(g.dP + o) = a0;
                *reinterpret_cast<f32x4*>(g.dVt + o) = a1;
            }
    }
#ifdef S2VT_AC_STAMP
    {
        const int nact = 4 * g.ncg;
        if (tid == 0 && (lin == 0 || lin == nact / 2 || lin == nact - 1)) {
            const int slot = lin == 0 ? 0 : (lin == nact - 1 ? 2 : 1);
#pragma unroll
            for (int i = 0; i < 16; ++i) atomicAdd(&ab_stamp_acc[slot * 16 + i], st_acc[i]);
        }
    }
#endif
}

// Behind the persistent launch, Tv > 5: dP[f][b][:] += sum_t de_t[f][b] w[:] (1 - tanh^2(hWa_t[b][:] + P[f][b][:])) and
// dV[f][b][:] += sum_t alpha_t[f][b] dctx_t[b][:], t descending -- the expressions and the order of the in-loop accumulation (and of
// attn.hip::attn_bwd_kernel), one workgroup per (row, frame) instead of one CU per row.
__global__ __launch_bounds__(256) void attn_dpdv_kernel(const float* P, const float* hWa, size_t hwa_tstride, const float* dcat, size_t dcat_tstride,
                                                        int ld_cat, const float* alpha, const float* deh, const float* w, int B, int H, int T, int Tv,
                                                        float* dP, float* dVt)
{
    const int b = blockIdx.x, f = blockIdx.y, q4 = threadIdx.x;
    if (4 * q4 >= H) return;
    const size_t o = ((size_t)f * B + b) * H + 4 * q4;
    const f32x4 pv = *reinterpret_cast<const f32x4*>(P + o), wh = *reinterpret_cast<const f32x4*>(w + 4 * q4);
    f32x4 ap = *reinterpret_cast<const f32x4*>(dP + o), av = *reinterpret_cast<const f32x4*>(dVt + o);
    for (int t0 = T - 1; t0 >= 0; t0 -= 4) {
        f32x4 hv[4], dn[4];
        float d[4], al[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 - u;
            hv[u] = dn[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            d[u] = al[u] = 0.f;
            if (t >= 0) {
                d[u] = deh[((size_t)t * Tv + f) * B + b];
                al[u] = alpha[((size_t)t * Tv + f) * B + b];
                if (t > 0) hv[u] = *reinterpret_cast<const f32x4*>(hWa + (size_t)t * hwa_tstride + (size_t)b * H + 4 * q4);
                dn[u] = *reinterpret_cast<const f32x4*>(dcat + (size_t)t * dcat_tstride + (size_t)b * ld_cat + H + 4 * q4);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (t0 - u >= 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float Tn = dm_tanhf(hv[u][i] + pv[i]);
                    const float s_ = d[u] * wh[i] * (1.f - Tn * Tn);
                    ap[i] = s_ + ap[i];
                    av[i] = al[u] * dn[u][i] + av[i];
                }
            }
    }
    *reinterpret_cast<f32x4*>(dP + o) = ap;
    *reinterpret_cast<f32x4*>(dVt + o) = av;
}

struct ABCfg { int ng; void (*fn)(const AttnBwdChainKArgs); const char* name; };
const ABCfg kABCfg[] = {{8, attn_bwd_chain_kernel<8>, "attn_bchain(ng8)"}, {64, attn_bwd_chain_kernel<64>, "attn_bchain(ng64)"}};
constexpr int kNumABCfg = 2;
int ab_lds_bytes(int ng) { return (ng * 2 * 256 + 4 * 16 * 20 + 4 * 16 * 17 + 4 * 256 + ng * 16 + 3 * kABMaxTv) * 4; }

struct ABDev {
    std::once_flag once;
    bool ok = false;
    int per_cu[kNumABCfg] = {};
};
constexpr int kMaxDev = 32;
ABDev g_abdev[kMaxDev];
ABDev* abdev_state()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
    ABDev& d = g_abdev[dev];
    std::call_once(d.once, [&d] {
        bool ok = true;
        for (int i = 0; ok && i < kNumABCfg; ++i) {
            const int lds = ab_lds_bytes(kABCfg[i].ng);
            ok = hipFuncSetAttribute(reinterpret_cast<const void*>(kABCfg[i].fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
            int n = 0;
            if (ok && hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kABCfg[i].fn), 256, lds) == hipSuccess) d.per_cu[i] = n;
        }
        d.ok = ok;
    });
    return &d;
}
int ab_cfg(int H) { return (H + 15) / 16 <= 8 ? 0 : 1; }

}  // namespace

bool attn_bwd_chain_eligible(int B, int H, int Tv)
{
    static const bool off = [] { const char* e = getenv("S2VT_ABCHAIN"); return e && e[0] == '0'; }();     // dev / test knob: per-step launches
    if (off || chain_persistent_disabled()) return false;
    ChainHost hst;
    ABDev* d = abdev_state();
    if (!chain_host(&hst) || !d || !d->ok) return false;
    if (!(B >= 1 && B <= 64 && H >= 16 && (H & 3) == 0 && H <= 1024 && Tv >= 1 && Tv <= kABMaxTv)) return false;
    const int ncg = (H + 15) / 16;
    if (B > 4 * ncg) return false;                               // a batch row per attention workgroup
    return (long)d->per_cu[ab_cfg(H)] * hst.num_cus >= 4L * ncg; // every ACTIVE workgroup resident at once
}

void attn_bwd_chain_scratch(int H, size_t* img_floats, size_t* ex_floats, size_t* row_floats, size_t* sync_bytes)
{
    const int ng = kABCfg[ab_cfg(H)].ng, ncg = (H + 15) / 16;
    *img_floats = (size_t)(8 + 1) * 4 * ng * 256;               // 2 parities x 4 gate images of dz + the dhWa image
    *ex_floats = (size_t)ncg * 4 * 8 * 256;
    *row_floats = (size_t)64 * H;
    *sync_bytes = 3 * kChainSyncBytes + (size_t)ncg * 128;
}

hipError_t launch_attn_bwd_chain(const AttnBwdChainLaunch& a, hipStream_t st)
{
    if (!attn_bwd_chain_eligible(a.B, a.H, a.Tv)) return hipErrorInvalidValue;
    if (a.T <= 0) return hipSuccess;
    if (a.Tv > kABRegFrames && !a.deh) return hipErrorInvalidValue;
    const auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    if (!al16(a.W3) || (a.ldw & 3) || !al16(a.Wa) || !al16(a.P) || !al16(a.Vt) || !al16(a.w) || !al16(a.hWa) || !al16(a.dhWa) || !al16(a.dP) || !al16(a.dVt) ||
        !al16(a.dcat) || (a.ld_cat & 3) || (a.dcat_tstride & 3) || !al16(a.img) || !al16(a.ex) || !al16(a.dctxs) || (a.hwa_tstride & 3) || (a.dhwa_tstride & 3))
        return hipErrorInvalidValue;
    ChainHost hst;
    if (!chain_host(&hst)) return hipErrorInvalidValue;
    const int ci = ab_cfg(a.H), ng = kABCfg[ci].ng, ncg = (a.H + 15) / 16;
    AttnBwdChainKArgs k;
    std::memset(&k, 0, sizeof(k));
    k.W3 = a.W3; k.ldw = a.ldw; k.Wa = a.Wa; k.ldwa = a.ldwa; k.gates = a.gates; k.gates_tstride = a.gates_tstride; k.C = a.C; k.state_tstride = a.state_tstride;
    k.dcat = a.dcat; k.dcat_tstride = a.dcat_tstride; k.ld_cat = a.ld_cat; k.dZ = a.dZ; k.dz_tstride = a.dz_tstride;
    k.dctx_hist = a.dcat; k.deh = a.deh;
    k.hWa = a.hWa; k.hwa_tstride = a.hwa_tstride; k.P = a.P; k.Vt = a.Vt; k.w = a.w; k.alpha = a.alpha;
    k.reg_coef = a.reg_coef; k.asum = a.asum; k.reg_m = a.reg_m; k.dhWa = a.dhWa; k.dhwa_tstride = a.dhwa_tstride; k.dP = a.dP; k.dVt = a.dVt; k.dw = a.dw;
    k.B = a.B; k.H = a.H; k.T = a.T; k.Tv = a.Tv;
    k.keep = a.keep; k.seed_lo = a.seed_lo; k.seed_hi = a.seed_hi; k.drop_code0 = a.drop_code0; k.video_id = a.video_id; k.sample_id = a.sample_id;
    size_t imgf, exf, rowf, syncb;
    attn_bwd_chain_scratch(a.H, &imgf, &exf, &rowf, &syncb);
    k.img = a.img; k.qimg = a.img + (size_t)8 * 4 * ng * 256; k.ex = a.ex; k.dctxs = a.dctxs; k.sync = a.sync;
    k.status = hst.status_dev; k.fault = hst.fault; k.spin_limit = hst.spin_limit; k.ncg = ncg;
    ChainLaunchOrder order;                                    // one persistent grid at a time per process
    {
        hipError_t we = order.before(st, hst.device);
        if (we != hipSuccess) return we;
    }
    ZeroList z;
    z.add(a.sync, syncb); z.add(a.img, imgf * 4);              // (rows >= B and k >= H of the images must read as zeros)
    hipError_t e = launch_zero_regions(z, st);
    if (e != hipSuccess) return e;
    e = chain_gate_zeroed(st);                                 // (gated overlap: a side stream may poll the counters from here on)
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)((ncg + 7) / 8 * 32));
    const double flops = (2.0 * a.B * (double)(2 * a.H) * 4.0 * a.H + 2.0 * a.B * (double)a.H * a.H) * a.T;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool prof = prof_wants(10, ci);
    if (prof) {
        hipError_t pe = prof_events(&e0, &e1);
        if (pe != hipSuccess) return pe;
        (void)hipEventRecord(e0, st);
    }
    hipLaunchKernelGGL(kABCfg[ci].fn, grid, dim3(256), ab_lds_bytes(ng), st, k);
    if (prof) {
        (void)hipEventRecord(e1, st);
        prof_record(10, ci, kABCfg[ci].name, flops, e0, e1);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = chain_gate_launched(a.sync, 4u * (unsigned)ncg);       // every active workgroup arrives at the dz hand-off once per iteration
    if (e != hipSuccess) return e;
    e = order.after(st, hst.device);
    if (e != hipSuccess) return e;
    if (a.Tv > kABRegFrames) {                                 // the accumulation of dP / dV the kernel left out (see there)
        hipLaunchKernelGGL(attn_dpdv_kernel, dim3((unsigned)a.B, (unsigned)a.Tv), dim3(256), 0, st, a.P, a.hWa, a.hwa_tstride, a.dcat, a.dcat_tstride,
                           a.ld_cat, a.alpha, a.deh, a.w, a.B, a.H, a.T, a.Tv, a.dP, a.dVt);
        e = hipGetLastError();
    }
    return e;
}

}  // namespace s2vt

#ifdef S2VT_AC_STAMP
// dev build only: per-phase clock sums of attn_bwd_chain_kernel (3 workgroups x 16 phases), read and reset (tools/ab_stamp.py)
extern "C" int s2vt_ab_stamp_read(unsigned long long* out48)
{
    if (hipDeviceSynchronize() != hipSuccess) return -4;
    if (hipMemcpyFromSymbol(out48, HIP_SYMBOL(s2vt::ab_stamp_acc), 48 * sizeof(unsigned long long)) != hipSuccess) return -4;
    unsigned long long z[48] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(s2vt::ab_stamp_acc), z, sizeof(z)) == hipSuccess ? 0 : -4;
}
#endif
