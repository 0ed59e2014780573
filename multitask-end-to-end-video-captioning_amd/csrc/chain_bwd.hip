// chain_bwd.hip -- the BACKWARD recurrence of a BasicLSTMCell in one persistent launch: kernels,
// configurations, eligibility and launcher.  Shares the hand-off helpers and the launch state with chain.hip (chain_common.h).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "chain_common.h"

namespace s2vt {

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// The BACKWARD recurrence of a BasicLSTMCell (back-propagation through the unroll of tf_s2vt.py:113-153; tf.gradients,
// reinforcement_multisampling_tf_s2vt.py:650) in one launch.  Step t (T-1 down to 0), per row m and unit u:
//     dh = dext_t[m,u] (through the DropoutWrapper mask) + sum_{g,k} dz_{t+1}[m, gH + k] * Whh[u, gH + k]
//     dz_t[m, gH + u] = pointwise(dh, dc, gates_t, c_t, c_{t-1});   dc <- dc_t * sf
// Per-step launches spent ~25 us on this at M = 64 whatever M is (a 7 us pointwise launch + 4-16 split-K slabs of a skinny
// product at 17-36 % of the matrix peak, dz and the slabs through HBM).  Here, as in the forward kernel above, nothing but
// the per-step operands moves:
//   * workgroup (j, g) owns 16 hidden units (output columns) and ONE gate's quarter of the reduction: its [H x 16] slice
//     Whh[16j .. 16j+15, gH .. gH+H) (64 KB) is gathered into LDS once, in MFMA B-fragment order;
//   * dz_{t+1} crosses the chip as four per-gate images in A-fragment order (write-through 16-byte stores, sc1 loads
//     straight into a register ring): a CU streams M*H*4 bytes per step, what the forward streams;
//   * the four gate partials of a unit group meet through a 4 KB-per-workgroup exchange among the FOUR workgroups (j, 0..3)
//     (same blockIdx % 8: one XCD under round-robin placement -- speed only): a cluster counter, not a grid-wide wait;
//     workgroup (j, g) then finishes row tiles g*TMW .. of its 16 units: one (row, unit) per thread, dc stays in a register
//     for all T steps, the next step's gates / states / upstream gradient are already in registers (prefetched under the MFMAs);
//   * ONE grid-wide hand-off per step (dz_t images), the form of the forward kernel.
// The reduction is order-free (gradients; compared with float64 autograd within tolerance): two accumulators
// per tile break the dependent MFMA chain, the gate partials are summed in gate order.
struct BwdChainArgs {
    const float* W; int ldw; int kw0;                  // cell matrix [*, 4H]; Whh[u][c] = W[(kw0 + u) * ldw + c]
    const float* gates; size_t gates_tstride;          // activated gates [T][M][4H] (si | tj | sf | so) of the forward pass
    const float* C; size_t state_tstride;              // cell states [T+1][M][H]: c_{t-1} = slot t, c_t = slot t + 1
    const float* dext; size_t dext_tstride; int ld_ext; int dext_t0;   // d loss / d out_t for t >= dext_t0 at dext + (t - dext_t0) * tstride (rows ld_ext apart); NULL = none
    float* dZ; size_t dz_tstride;                      // [T][M][4H] pre-activation gradients (what the weight-gradient contractions read)
    int M, H, T;
    float keep; uint32_t seed_lo, seed_hi, drop_code0; // DropoutWrapper of `out`: code = drop_code0 + t
    const int32_t* video_id; const int32_t* sample_id;
    float* img;                                        // 2 x 4 gate images of dz in A-fragment order
    float* ex;                                         // [unit groups][4 gates][row tiles][256] partial tiles
    unsigned* sync;                                    // grid counters (kChainSyncBytes) then one 128-byte line per unit group
    unsigned* status; unsigned* fault; unsigned spin_limit;
    int ncg;                                           // unit groups = ceil(H / (16 NC))
    int tpp, img_tiles;                                // row tiles of a row part; row tiles of one image (= parts * tpp)
    const int32_t* perm; const int32_t* nlive;         // LIVE kernels: virtual row -> row of the arrays; live virtual rows per step
};

// NC = 16-unit column tiles per workgroup.  NC = 1: workgroup (j, g) takes every row tile (4 TMW of them).  NC = 2 (M > 256
// rows): workgroup (j, g, part) owns 32 units (128 KB of Whh in LDS) and HALF the row tiles (g.tpp per part), so a CU
// streams half of its gate's dz image per step for the same number of MFMAs -- the two-part form of the forward kernel.
template <int NG, int TMW, int NC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_bwd_chain_kernel(const BwdChainArgs g)
{
    constexpr int ZS = 20;
    constexpr int PS = NC * TMW;                               // (row tile, column tile) slots one workgroup finishes per step, at most
#ifdef S2VT_BCHAIN_PRIO
    __builtin_amdgcn_s_setprio(S2VT_BCHAIN_PRIO);              // (dev) beside a co-resident contraction (train.hip, gated overlap): issue the recurrence's instructions first
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                                          // [NG][NC][64 lanes][4]: B fragments of this workgroup's slice
    const int tid = threadIdx.x, lane = tid & 63;
    const int pwave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* zb = smem + NG * NC * 256 + pwave * (16 * ZS);      // per-wave transpose tile
    float* dzl = smem + NG * NC * 256 + 4 * 16 * ZS;           // [4 gates][16 rows][17]: dz of one finished tile, regrouped for the image stores
    const int l15 = lane & 15, lq = lane >> 4;
    const int H = g.H, M = g.M, T = g.T;
    // workgroup id -> (unit group, gate, row part): the four gates of a (unit group, part) share blockIdx % 8
    const int wg = (int)blockIdx.x;
    const int gate = (wg >> 3) & 3;
    const int part = NC == 2 ? (wg >> 5) & 1 : 0;
    const int jj = (wg >> (NC == 2 ? 6 : 5)) * 8 + (wg & 7);
    if (jj >= g.ncg) return;                                   // (grid padded to whole groups of 8: these never take part)
    const int u0 = jj * 16 * NC;
    const int tpp = g.tpp;                                     // row tiles of a part
    const int NT = g.img_tiles;                                // row tiles of an image (= parts * tpp)
    const size_t img_floats = (size_t)NT * NG * 256;           // one gate image

    // ---- this workgroup's slice of Whh -> LDS, once.  B[k][n] = Whh[u0 + n][gate * H + k]; element (k, n) of column tile
    // c = n / 16 goes to group k / 16, tile c, lane (k % 4) * 16 + n % 16, component (k % 16) / 4.
    for (int idx = tid; idx < 16 * NC * NG * 4; idx += 256) {
        const int k4 = idx % (NG * 4), n = idx / (NG * 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (u0 + n < H && 4 * k4 < H) v = *reinterpret_cast<const f32x4*>(g.W + (size_t)(g.kw0 + u0 + n) * g.ldw + (size_t)gate * H + 4 * k4);
        const int grp = k4 >> 2, e = k4 & 3, c = n >> 4, nn = n & 15;
#pragma unroll
        for (int i = 0; i < 4; ++i) Wl[(((grp * NC + c) * 64 + i * 16 + nn) << 2) + e] = v[i];
    }

    // ---- the (row, unit) this thread finishes at every step, per slot s = gate * PSr + i of the part's tpp * NC (row tile,
    // column tile) pairs (PSr = slots per workgroup, the four gate workgroups share them)
    const int pr = tid >> 4, pn = tid & 15;                    // row within the tile, unit within the column tile
    const int nslots = tpp * NC, psr = (nslots + 3) >> 2;
    float dc_reg[PS], cnew[PS];
    uint32_t vid[PS], sid[PS];
    bool pok[PS];
    int pm[PS], pu[PS], ptile[PS], pc[PS];
#pragma unroll
    for (int i = 0; i < PS; ++i) {
        const int slot = gate * psr + i;
        ptile[i] = slot / NC; pc[i] = slot % NC;              // row tile within the part, column tile
        pm[i] = (part * tpp + ptile[i]) * 16 + pr;
        pu[i] = u0 + pc[i] * 16 + pn;
        pok[i] = i < psr && slot < nslots && pm[i] < M && pu[i] < H;
        dc_reg[i] = 0.0f;
        cnew[i] = pok[i] ? g.C[(size_t)T * g.state_tstride + (size_t)pm[i] * H + pu[i]] : 0.0f;      // c_{T-1}
        vid[i] = (g.keep < 1.0f && pok[i]) ? (uint32_t)g.video_id[pm[i]] : 0u;
        sid[i] = (g.keep < 1.0f && pok[i]) ? (uint32_t)g.sample_id[pm[i]] : 0u;
    }
    const int cluster = jj * NC + part;
    gu32* const ccount = (gu32*)g.sync + (kChainSyncBytes / 4) + cluster * 32;
    const __amdgpu_buffer_rsrc_t rsEx = __builtin_amdgcn_make_buffer_rsrc(g.ex, 0, g.ncg * NC * 4 * tpp * NC * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsImg = __builtin_amdgcn_make_buffer_rsrc(g.img, 0, (int)(8 * img_floats * 4), 0x00020000);
    GridSync gs{(gu32*)g.sync, g.status, g.fault, g.spin_limit, g.ncg, false, 4u * NC};
    bool wok[TMW];                                             // MFMA side: row tile pwave * TMW + i of the part holds rows of the problem
    int voff[TMW];
    const int tb = part * tpp + pwave * TMW;                   // first row tile of this wave
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
        wok[i] = pwave * TMW + i < tpp && (tb + i) * 16 < M;
        voff[i] = wok[i] ? lane * 16 : (int)0x80000000u;
    }
    __syncthreads();

    float sg[PS][4], cprev[PS], dx[PS];
    auto load_step = [&](int t) __attribute__((always_inline)) {                // operands of step t's pointwise part (independent of the recurrence)
#pragma unroll
        for (int i = 0; i < PS; ++i) {
            const float* gp = g.gates + (size_t)t * g.gates_tstride + (size_t)pm[i] * 4 * H + pu[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) sg[i][q] = pok[i] ? gp[(size_t)q * H] : 0.0f;
            cprev[i] = pok[i] ? g.C[(size_t)t * g.state_tstride + (size_t)pm[i] * H + pu[i]] : 0.0f;
            dx[i] = (pok[i] && g.dext && t >= g.dext_t0) ? g.dext[(size_t)(t - g.dext_t0) * g.dext_tstride + (size_t)pm[i] * g.ld_ext + pu[i]] : 0.0f;
        }
    };
    load_step(T - 1);

    unsigned arrival = 0;
    for (int t = T - 1; t >= 0; --t) {
        float dh[PS];
#pragma unroll
        for (int i = 0; i < PS; ++i) dh[i] = 0.0f;
        if (t < T - 1) {
            // ---- dz_{t+1}[:, gate block] @ slice^T for this wave's row tiles: A fragments straight into registers
            gs.wait_all(arrival, pwave, lane);
            const float* acur = g.img + (size_t)((t + 1) & 1) * 4 * img_floats + (size_t)gate * img_floats;
            const __amdgpu_buffer_rsrc_t rsA =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(acur + (size_t)tb * NG * 256), 0, TMW * NG * 1024, 0x00020000);
            constexpr int RING0 = TMW == 1 ? 32 : 40 / TMW;
            constexpr int RING = NG < RING0 ? NG : RING0;
            f32x4 a[RING][TMW];
            f32x4 acc[2][NC][TMW];
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int i = 0; i < TMW; ++i) { acc[0][c][i] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1][c][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            static_for<0, RING>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                static_for<0, TMW>([&](auto i_) { constexpr int i = decltype(i_)::value; a[j][i] = bload16_sc1(rsA, voff[i], (i * NG + j) * 1024); });
            });
            __builtin_amdgcn_sched_barrier(0);
            const f32x4* bl = reinterpret_cast<const f32x4*>(Wl) + lane;
            constexpr int PB = NG < 4 ? NG : 4;
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
                            acc[e & 1][c][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j % RING][i][e], bj4[c][e], acc[e & 1][c][i], 0, 0, 0);
                        });
                    });
                });
                if constexpr (j + RING < NG) {
                    __builtin_amdgcn_sched_barrier(0);
                    static_for<0, TMW>([&](auto i_) { constexpr int i = decltype(i_)::value; a[j % RING][i] = bload16_sc1(rsA, voff[i], (i * NG + j + RING) * 1024); });
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            // ---- partial tiles -> the cluster's exchange [gate][row tile of the part][column tile] (row-major 16 x 16, one
            // write-through 16-byte store per lane)
            const size_t exc = (size_t)cluster * 4 * nslots;          // tiles of this cluster's exchange before its own
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) zb[(lq * 4 + r) * ZS + l15] = acc[0][c][i][r] + acc[1][c][i][r];
                    __builtin_amdgcn_wave_barrier();
                    const f32x4 row = *reinterpret_cast<const f32x4*>(zb + (lane >> 2) * ZS + (lane & 3) * 4);
                    __builtin_amdgcn_wave_barrier();
                    const int tl = pwave * TMW + i;                    // row tile within the part
                    if (tl < tpp)
                        bstore16_sc1(rsEx, __builtin_bit_cast(u32x4v, row), (int)(((exc + (size_t)gate * nslots + (size_t)tl * NC + c) * 256 + lane * 4) * 4), 0);
                }
            gs.arrive_one(ccount, tid);
            gs.wait_one(ccount, 4u * (arrival + 1u), pwave, lane);
            ++arrival;
#pragma unroll
            for (int i = 0; i < PS; ++i) {
                const int slot = gate * psr + i;
                float s_ = 0.0f;
                if (i < psr && slot < nslots) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        s_ += __uint_as_float(__hip_atomic_load((const gu32*)(g.ex + (exc + (size_t)q * nslots + slot) * 256 + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                }
                dh[i] = s_;
            }
        }
        // ---- BasicLSTMCell backward pointwise (the expressions of lstm_bwd_pointwise_kernel), one (row, unit) per thread and slot
        float dzv[PS][4];
#pragma unroll
        for (int i = 0; i < PS; ++i) {
            float d = dx[i];
            if (g.keep < 1.0f && g.dext && t >= g.dext_t0)
                d = (d / g.keep) * dropout_keep01(g.seed_lo, g.seed_hi, vid[i], sid[i], g.drop_code0 + (uint32_t)t, (uint32_t)pu[i], g.keep);
            const float dht = dh[i] + d;
            const float si = sg[i][0], tj = sg[i][1], sf = sg[i][2], so = sg[i][3];
            const float tc = dm_tanhf(cnew[i]);
            const float dc = dht * so * (1.f - tc * tc) + dc_reg[i];
            dzv[i][0] = dc * tj * si * (1.f - si);
            dzv[i][1] = dc * si * (1.f - tj * tj);
            dzv[i][2] = dc * cprev[i] * sf * (1.f - sf);
            dzv[i][3] = dht * tc * so * (1.f - so);
            dc_reg[i] = dc * sf;
            cnew[i] = cprev[i];                                  // c_{t-1} is the next step's c_t
        }
        if (t > 0) {
            // ---- dz_t -> the four gate images of the other parity, regrouped through LDS so that every thread writes ONE
            // 16-byte fragment slot: thread (gate q = tid / 64, slot L = tid % 64) takes row L % 16, units L / 16 + 4e
            const size_t inext = (size_t)(t & 1) * 4 * img_floats;
#pragma unroll
            for (int i = 0; i < PS; ++i) {
                const int slot = gate * psr + i;
                if (!(i < psr && slot < nslots)) continue;       // (uniform over the workgroup)
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 4; ++q) dzl[(q * 16 + pr) * 17 + pn] = pok[i] ? dzv[i][q] : 0.0f;
                __syncthreads();
                const int q = tid >> 6, L = tid & 63, r = L & 15, kq = L >> 4;
                u32x4v w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = __float_as_uint(dzl[(q * 16 + r) * 17 + kq + 4 * e]);
                const size_t dst = inext + (size_t)q * img_floats + ((size_t)((part * tpp + ptile[i]) * NG + jj * NC + pc[i]) * 64 + L) * 4;
                bstore16_sc1(rsImg, w, (int)(dst * 4), 0);
            }
            gs.arrive(tid);
        }
        // ---- history: dZ[t] (read by the weight-gradient contractions after the launch), then the next step's operands
#pragma unroll
        for (int i = 0; i < PS; ++i) {
            if (!pok[i]) continue;
            float* zp = g.dZ + (size_t)t * g.dz_tstride + (size_t)pm[i] * 4 * H + pu[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) zp[(size_t)q * H] = dzv[i][q];
        }
        if (t > 0) load_step(t - 1);
    }
}

// The backward recurrence above 256 rows in the register-weights form (the construction of lstm_chain4_kernel): workgroup
// (j, gate, part) = 64 output units x one gate's quarter of the reduction x a QUARTER of the row tiles (grid 16 x 4 x 4 = 256
// at H = 1000); wave w keeps the B fragments of column tile w (units 64j + 16w .. +15, all H of the gate's k) in registers
// for the whole launch; the part's slice of the gate's dz image goes global -> LDS once per CU (LDS-DMA ring) and every
// wave reads the same A fragments from it: TPP row tiles x 1 column tile per wave (balanced), a quarter of the image per CU.
// Exchange, pointwise part and hand-off are those of the kernel above with 4 column tiles per workgroup.
template <int NG, int TPP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_bwd_chain4_kernel(const BwdChainArgs g)
{
    constexpr int NC = 4;                                      // (name of the kernel above: column tiles per workgroup)
    constexpr int CG = 8, NCH = NG / CG, NBUF = 3;             // k-groups per chunk, chunks per step, LDS chunk buffers
    constexpr int CHF = TPP * CG * 256;                        // floats per chunk buffer
    constexpr int DPW = TPP * CG / 4;                          // DMA instructions per wave per chunk
    static_assert(NG % CG == 0 && NCH >= NBUF && (TPP * CG) % 4 == 0, "chunking");
    constexpr int ZS = 20;
    constexpr int PS = TPP;                                    // (row tile, column tile) slots one workgroup finishes per step: tpp * 4 / 4 gates
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ab = smem;                                          // [NBUF][CG][TPP][64 lanes][4]: the A ring
    const int tid = threadIdx.x, lane = tid & 63;
    const int pwave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* zb = smem + NBUF * CHF + pwave * (16 * ZS);         // per-wave transpose tile
    float* dzl = smem + NBUF * CHF + 4 * 16 * ZS;              // [4 gates][16 rows][17]: dz of one finished tile, regrouped for the image stores
    const int l15 = lane & 15, lq = lane >> 4;
    const int H = g.H, M = g.M, T = g.T;
    // workgroup id -> (unit group, gate, row part): the four gates of a (unit group, part) share blockIdx % 8
    const int wg = (int)blockIdx.x;
    const int gate = (wg >> 3) & 3;
    const int part = (wg >> 5) & 3;
    const int jj = (wg >> 7) * 8 + (wg & 7);
    if (jj >= g.ncg) return;                                   // (grid padded to whole groups of 8: these never take part)
    const int u0 = jj * 16 * NC;
    const int tpp = g.tpp;                                     // row tiles of a part
    const int NT = g.img_tiles;                                // row tiles of an image (= parts * tpp)
    const size_t img_floats = (size_t)NT * NG * 256;           // one gate image

    // ---- this wave's column tile of the slice -> registers, once: k-step s holds B[k = 4s + lq][n = l15] = Whh[u0 + 16 w + l15][gate H + k]
    float breg[4 * NG];
    {
        const int un = u0 + 16 * pwave + l15;
#pragma unroll
        for (int s_ = 0; s_ < 4 * NG; ++s_) {
            const int kk = 4 * s_ + lq;
            breg[s_] = (un < H && kk < H) ? g.W[(size_t)(g.kw0 + un) * g.ldw + (size_t)gate * H + kk] : 0.0f;
        }
    }

    // ---- the (row, unit) this thread finishes at every step, per slot s = gate * PSr + i of the part's tpp * NC (row tile,
    // column tile) pairs (PSr = slots per workgroup, the four gate workgroups share them)
    const int pr = tid >> 4, pn = tid & 15;                    // row within the tile, unit within the column tile
    const int nslots = tpp * NC, psr = (nslots + 3) >> 2;
    float dc_reg[PS], cnew[PS];
    uint32_t vid[PS], sid[PS];
    bool pok[PS];
    int pm[PS], pu[PS], ptile[PS], pc[PS];
#pragma unroll
    for (int i = 0; i < PS; ++i) {
        const int slot = gate * psr + i;
        ptile[i] = slot / NC; pc[i] = slot % NC;              // row tile within the part, column tile
        pm[i] = (part * tpp + ptile[i]) * 16 + pr;
        pu[i] = u0 + pc[i] * 16 + pn;
        pok[i] = i < psr && slot < nslots && pm[i] < M && pu[i] < H;
        dc_reg[i] = 0.0f;
        cnew[i] = pok[i] ? g.C[(size_t)T * g.state_tstride + (size_t)pm[i] * H + pu[i]] : 0.0f;      // c_{T-1}
        vid[i] = (g.keep < 1.0f && pok[i]) ? (uint32_t)g.video_id[pm[i]] : 0u;
        sid[i] = (g.keep < 1.0f && pok[i]) ? (uint32_t)g.sample_id[pm[i]] : 0u;
    }
    const int cluster = jj * NC + part;
    gu32* const ccount = (gu32*)g.sync + (kChainSyncBytes / 4) + cluster * 32;
    const __amdgpu_buffer_rsrc_t rsEx = __builtin_amdgcn_make_buffer_rsrc(g.ex, 0, g.ncg * NC * 4 * tpp * NC * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsImg = __builtin_amdgcn_make_buffer_rsrc(g.img, 0, (int)(8 * img_floats * 4), 0x00020000);
    GridSync gs{(gu32*)g.sync, g.status, g.fault, g.spin_limit, g.ncg, false, 4u * NC};
    const int tb = part * tpp;                                 // first row tile of this workgroup (every wave multiplies all tpp of them)
    __syncthreads();

    float sg[PS][4], cprev[PS], dx[PS];
    auto load_step = [&](int t) __attribute__((always_inline)) {                // operands of step t's pointwise part (independent of the recurrence)
#pragma unroll
        for (int i = 0; i < PS; ++i) {
            const float* gp = g.gates + (size_t)t * g.gates_tstride + (size_t)pm[i] * 4 * H + pu[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) sg[i][q] = pok[i] ? gp[(size_t)q * H] : 0.0f;
            cprev[i] = pok[i] ? g.C[(size_t)t * g.state_tstride + (size_t)pm[i] * H + pu[i]] : 0.0f;
            dx[i] = (pok[i] && g.dext && t >= g.dext_t0) ? g.dext[(size_t)(t - g.dext_t0) * g.dext_tstride + (size_t)pm[i] * g.ld_ext + pu[i]] : 0.0f;
        }
    };
    load_step(T - 1);

    unsigned arrival = 0;
    for (int t = T - 1; t >= 0; --t) {
        float dh[PS];
#pragma unroll
        for (int i = 0; i < PS; ++i) dh[i] = 0.0f;
        if (t < T - 1) {
            // ---- dz_{t+1}[:, gate block] @ slice^T for this wave's row tiles: A fragments straight into registers
            gs.wait_all(arrival, pwave, lane);
            const float* acur = g.img + (size_t)((t + 1) & 1) * 4 * img_floats + (size_t)gate * img_floats;
            const __amdgpu_buffer_rsrc_t rsA =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(acur + (size_t)tb * NG * 256), 0, TPP * NG * 1024, 0x00020000);
            // chunk c = groups c*CG ..; piece p = gq * TPP + i; wave w issues pieces w, w + 4, .. (static: CG * TPP / 4 each)
            auto issue_chunk = [&](auto c_) __attribute__((always_inline)) {
                constexpr int c = decltype(c_)::value;
                float* dstb = Ab + (c % NBUF) * CHF;
                static_for<0, DPW>([&](auto q_) {
                    constexpr int q = decltype(q_)::value;
                    const int p = pwave + 4 * q;
                    const int gq = p / TPP, i = p % TPP;
                    const int vo = (tb + i) * 16 < M ? lane * 16 : (int)0x80000000u;          // row tiles beyond the problem: zeros
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr4)(dstb + p * 256), 16, vo, (i * NG + c * CG + gq) * 1024, 0, 16);   // aux 16 = sc1
                });
            };
            f32x4 acc[TPP];
#pragma unroll
            for (int i = 0; i < TPP; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            static_for<0, NBUF - 1>([&](auto c_) { issue_chunk(c_); });
            static_for<0, NCH>([&](auto c_) {
                constexpr int c = decltype(c_)::value;
                constexpr int issued = c + NBUF - 1 < NCH ? c + NBUF - 1 : NCH;
                constexpr int later = issued - (c + 1);
                static_assert(later * DPW <= 63, "vmcnt range");
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(later * DPW) : "memory");
                __syncthreads();
                if constexpr (c + NBUF - 1 < NCH) issue_chunk(std::integral_constant<int, c + NBUF - 1>{});
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
            __syncthreads();                                          // (everybody is done with the ring before the exchange tiles reuse zb / the next step's DMA)
            // ---- partial tiles -> the cluster's exchange [gate][row tile of the part][column tile] (row-major 16 x 16, one
            // write-through 16-byte store per lane)
            const size_t exc = (size_t)cluster * 4 * nslots;          // tiles of this cluster's exchange before its own
#pragma unroll
            for (int i = 0; i < TPP; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) zb[(lq * 4 + r) * ZS + l15] = acc[i][r];
                __builtin_amdgcn_wave_barrier();
                const f32x4 row = *reinterpret_cast<const f32x4*>(zb + (lane >> 2) * ZS + (lane & 3) * 4);
                __builtin_amdgcn_wave_barrier();
                if (i < tpp)
                    bstore16_sc1(rsEx, __builtin_bit_cast(u32x4v, row), (int)(((exc + (size_t)gate * nslots + (size_t)i * NC + pwave) * 256 + lane * 4) * 4), 0);
            }
            gs.arrive_one(ccount, tid);
            gs.wait_one(ccount, 4u * (arrival + 1u), pwave, lane);
            ++arrival;
#pragma unroll
            for (int i = 0; i < PS; ++i) {
                const int slot = gate * psr + i;
                float s_ = 0.0f;
                if (i < psr && slot < nslots) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        s_ += __uint_as_float(__hip_atomic_load((const gu32*)(g.ex + (exc + (size_t)q * nslots + slot) * 256 + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                }
                dh[i] = s_;
            }
        }
        // ---- BasicLSTMCell backward pointwise (the expressions of lstm_bwd_pointwise_kernel), one (row, unit) per thread and slot
        float dzv[PS][4];
#pragma unroll
        for (int i = 0; i < PS; ++i) {
            float d = dx[i];
            if (g.keep < 1.0f && g.dext && t >= g.dext_t0)
                d = (d / g.keep) * dropout_keep01(g.seed_lo, g.seed_hi, vid[i], sid[i], g.drop_code0 + (uint32_t)t, (uint32_t)pu[i], g.keep);
            const float dht = dh[i] + d;
            const float si = sg[i][0], tj = sg[i][1], sf = sg[i][2], so = sg[i][3];
            const float tc = dm_tanhf(cnew[i]);
            const float dc = dht * so * (1.f - tc * tc) + dc_reg[i];
            dzv[i][0] = dc * tj * si * (1.f - si);
            dzv[i][1] = dc * si * (1.f - tj * tj);
            dzv[i][2] = dc * cprev[i] * sf * (1.f - sf);
            dzv[i][3] = dht * tc * so * (1.f - so);
            dc_reg[i] = dc * sf;
            cnew[i] = cprev[i];                                  // c_{t-1} is the next step's c_t
        }
        if (t > 0) {
            // ---- dz_t -> the four gate images of the other parity, regrouped through LDS so that every thread writes ONE
            // 16-byte fragment slot: thread (gate q = tid / 64, slot L = tid % 64) takes row L % 16, units L / 16 + 4e
            const size_t inext = (size_t)(t & 1) * 4 * img_floats;
#pragma unroll
            for (int i = 0; i < PS; ++i) {
                const int slot = gate * psr + i;
                if (!(i < psr && slot < nslots)) continue;       // (uniform over the workgroup)
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 4; ++q) dzl[(q * 16 + pr) * 17 + pn] = pok[i] ? dzv[i][q] : 0.0f;
                __syncthreads();
                const int q = tid >> 6, L = tid & 63, r = L & 15, kq = L >> 4;
                u32x4v w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = __float_as_uint(dzl[(q * 16 + r) * 17 + kq + 4 * e]);
                const size_t dst = inext + (size_t)q * img_floats + ((size_t)((part * tpp + ptile[i]) * NG + jj * NC + pc[i]) * 64 + L) * 4;
                bstore16_sc1(rsImg, w, (int)(dst * 4), 0);
            }
            gs.arrive(tid);
        }
        // ---- history: dZ[t] (read by the weight-gradient contractions after the launch), then the next step's operands
#pragma unroll
        for (int i = 0; i < PS; ++i) {
            if (!pok[i]) continue;
            float* zp = g.dZ + (size_t)t * g.dz_tstride + (size_t)pm[i] * 4 * H + pu[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) zp[(size_t)q * H] = dzv[i][q];
        }
        if (t > 0) load_step(t - 1);
    }
}

// The same with LIVE ROWS (round 4): the rows are VIRTUAL, sorted by caption length (row v of the images is row g.perm[v] of the caller's arrays), the
// tiles of a part interleaved (tile i of part p = image tile 4 i + p), and step t runs only the row tiles that hold one of the
// first g.nlive[t] virtual rows -- the form of lstm_chain4_kernel<.., true> (chain.hip).  A tile's steps form a suffix-free prefix
// 0 .. len-1 of the unroll, so walking backwards a tile JOINS once and then stays: it joins with dc = 0 and a zero dz image (the
// launcher zeroes the images; a dead tile never writes its slots), exactly what the dense pass computes for it from the zero
// upstream gradients of the masked positions.  Workgroup (j, gate, part) then finishes column tile `gate` of every live row tile.
// (A kernel of its own rather than a flag of the one above: at 510 of 512 registers the dense form's allocation does not survive
// being generated from shared source -- 14 spills and 60 bytes of scratch when it was tried.)
template <int NG, int TPP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_bwd_chain4_live_kernel(const BwdChainArgs g)
{
    constexpr bool LIVE = true;
    constexpr int NC = 4;                                      // (name of the kernel above: column tiles per workgroup)
    constexpr int CG = 8, NCH = NG / CG, NBUF = 3;             // k-groups per chunk, chunks per step, LDS chunk buffers
    constexpr int CHF = TPP * CG * 256;                        // floats per chunk buffer
    static_assert(NG % CG == 0 && NCH >= NBUF && CG % 4 == 0, "chunking");
    constexpr int ZS = 20;
    constexpr int PS = TPP;                                    // (row tile, column tile) slots one workgroup finishes per step: tpp * 4 / 4 gates
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ab = smem;                                          // [NBUF][CG][TPP][64 lanes][4]: the A ring
    const int tid = threadIdx.x, lane = tid & 63;
    const int pwave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* zb = smem + NBUF * CHF + pwave * (16 * ZS);         // per-wave transpose tile
    float* dzl = smem + NBUF * CHF + 4 * 16 * ZS;              // [4 gates][16 rows][17]: dz of one finished tile, regrouped for the image stores
    const int l15 = lane & 15, lq = lane >> 4;
    const int H = g.H, M = g.M, T = g.T;
    // workgroup id -> (unit group, gate, row part): the four gates of a (unit group, part) share blockIdx % 8
    const int wg = (int)blockIdx.x;
    const int gate = (wg >> 3) & 3;
    const int part = (wg >> 5) & 3;
    const int jj = (wg >> 7) * 8 + (wg & 7);
    if (jj >= g.ncg) return;                                   // (grid padded to whole groups of 8: these never take part)
    const int u0 = jj * 16 * NC;
    const int tpp = g.tpp;                                     // row tiles of a part
    const int NT = g.img_tiles;                                // row tiles of an image (= parts * tpp)
    const size_t img_floats = (size_t)NT * NG * 256;           // one gate image
    constexpr int TS = LIVE ? 4 : 1;                           // image tiles between consecutive row tiles of this workgroup
    const int tile0 = LIVE ? part : part * tpp;                // image tile of this workgroup's row tile 0

    // ---- this wave's column tile of the slice -> registers, once: k-step s holds B[k = 4s + lq][n = l15] = Whh[u0 + 16 w + l15][gate H + k]
    float breg[4 * NG];
    {
        const int un = u0 + 16 * pwave + l15;
#pragma unroll
        for (int s_ = 0; s_ < 4 * NG; ++s_) {
            const int kk = 4 * s_ + lq;
            breg[s_] = (un < H && kk < H) ? g.W[(size_t)(g.kw0 + un) * g.ldw + (size_t)gate * H + kk] : 0.0f;
        }
    }

    // ---- the (row, unit) this thread finishes at every step, per slot of the part's tpp * NC (row tile, column tile) pairs.
    // Dense: slot = gate * PSr + i (PSr = slots per workgroup, the four gate workgroups share them in order); LIVE: row tile i,
    // column tile `gate` (balanced for any count of live tiles)
    const int pr = tid >> 4, pn = tid & 15;                    // row within the tile, unit within the column tile
    const int nslots = tpp * NC, psr = (nslots + 3) >> 2;
    float dc_reg[PS], cnew[PS];
    uint32_t vid[PS], sid[PS];
    bool pok[PS];
    int pm[PS], pu[PS], ptile[PS], pc[PS];
    auto slot_of = [&](int i) __attribute__((always_inline)) { return LIVE ? i * NC + gate : gate * psr + i; };
#pragma unroll
    for (int i = 0; i < PS; ++i) {
        const int slot = slot_of(i);
        ptile[i] = slot / NC; pc[i] = slot % NC;              // row tile within the part, column tile
        const int vrow = (tile0 + ptile[i] * TS) * 16 + pr;
        pu[i] = u0 + pc[i] * 16 + pn;
        pok[i] = i < psr && slot < nslots && vrow < M && pu[i] < H;
        pm[i] = vrow;
        if constexpr (LIVE) pm[i] = pok[i] ? (int)g.perm[vrow] : 0;
        dc_reg[i] = 0.0f;
        cnew[i] = pok[i] ? g.C[(size_t)T * g.state_tstride + (size_t)pm[i] * H + pu[i]] : 0.0f;      // c_{T-1}
        vid[i] = (g.keep < 1.0f && pok[i]) ? (uint32_t)g.video_id[pm[i]] : 0u;
        sid[i] = (g.keep < 1.0f && pok[i]) ? (uint32_t)g.sample_id[pm[i]] : 0u;
    }
    const int cluster = jj * NC + part;
    gu32* const ccount = (gu32*)g.sync + (kChainSyncBytes / 4) + cluster * 32;
    const __amdgpu_buffer_rsrc_t rsEx = __builtin_amdgcn_make_buffer_rsrc(g.ex, 0, g.ncg * NC * 4 * tpp * NC * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsImg = __builtin_amdgcn_make_buffer_rsrc(g.img, 0, (int)(8 * img_floats * 4), 0x00020000);
    GridSync gs{(gu32*)g.sync, g.status, g.fault, g.spin_limit, g.ncg, false, 4u * NC};
    __syncthreads();

    // live row tiles of this part at step t (dense: all)
    auto live_tiles = [&](int t) __attribute__((always_inline)) {
        if constexpr (!LIVE) return TPP;
        else {
            int n = g.nlive ? g.nlive[t] : M;
            n = n < M ? n : M;
            int nl = (((n + 15) >> 4) - part + 3) >> 2;
            nl = nl < 0 ? 0 : (nl > TPP ? TPP : nl);
            return __builtin_amdgcn_readfirstlane(nl);
        }
    };

    float sg[PS][4], cprev[PS], dx[PS];
    auto load_step = [&](int t, int nl) __attribute__((always_inline)) {        // operands of step t's pointwise part (independent of the recurrence)
#pragma unroll
        for (int i = 0; i < PS; ++i) {
            const bool ok = pok[i] && i < nl;
            const float* gp = g.gates + (size_t)t * g.gates_tstride + (size_t)pm[i] * 4 * H + pu[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) sg[i][q] = ok ? gp[(size_t)q * H] : 0.0f;
            cprev[i] = ok ? g.C[(size_t)t * g.state_tstride + (size_t)pm[i] * H + pu[i]] : 0.0f;
            dx[i] = (ok && g.dext && t >= g.dext_t0) ? g.dext[(size_t)(t - g.dext_t0) * g.dext_tstride + (size_t)pm[i] * g.ld_ext + pu[i]] : 0.0f;
            cnew[i] = ok ? g.C[(size_t)(t + 1) * g.state_tstride + (size_t)pm[i] * H + pu[i]] : 0.0f;      // (a tile that joins at this step has no c_t from the step before)
        }
    };
    const size_t exc = (size_t)cluster * 4 * nslots;              // tiles of this cluster's exchange before its own
    // dz_{t+1}[:, gate block] @ slice^T for NL row tiles, the partial tiles handed to the cluster's exchange: the only part of a
    // step that is instantiated per count of live tiles
    auto product = [&](auto nl_, const int t) __attribute__((always_inline)) {
        constexpr int NL = decltype(nl_)::value;
        constexpr int DPW = NL * CG / 4;                       // DMA instructions per wave per chunk
        const float* acur = g.img + (size_t)((t + 1) & 1) * 4 * img_floats + (size_t)gate * img_floats;
        const __amdgpu_buffer_rsrc_t rsA =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(acur + (size_t)tile0 * NG * 256), 0, ((TPP - 1) * TS + 1) * NG * 1024, 0x00020000);
        // chunk c = groups c*CG ..; piece p = gq * NL + i; wave w issues pieces w, w + 4, .. (static: CG * NL / 4 each)
        auto issue_chunk = [&](auto c_) __attribute__((always_inline)) {
            constexpr int c = decltype(c_)::value;
            float* dstb = Ab + (c % NBUF) * CHF;
            static_for<0, DPW>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                const int p = pwave + 4 * q;
                const int gq = p / NL, i = p % NL;
                const int vo = (tile0 + i * TS) * 16 < M ? lane * 16 : (int)0x80000000u;          // row tiles beyond the problem: zeros
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr4)(dstb + (gq * TPP + i) * 256), 16, vo, (i * TS * NG + c * CG + gq) * 1024, 0, 16);   // aux 16 = sc1
            });
        };
        f32x4 acc[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        static_for<0, NBUF - 1>([&](auto c_) { issue_chunk(c_); });
        static_for<0, NCH>([&](auto c_) {
            constexpr int c = decltype(c_)::value;
            constexpr int issued = c + NBUF - 1 < NCH ? c + NBUF - 1 : NCH;
            constexpr int later = issued - (c + 1);
            static_assert(later * DPW <= 63, "vmcnt range");
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(later * DPW) : "memory");
            __syncthreads();
            if constexpr (c + NBUF - 1 < NCH) issue_chunk(std::integral_constant<int, c + NBUF - 1>{});
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
        __syncthreads();                                          // (everybody is done with the ring before the exchange tiles reuse zb / the next step's DMA)
        // ---- partial tiles -> the cluster's exchange [gate][row tile of the part][column tile] (row-major 16 x 16, one
        // write-through 16-byte store per lane)
#pragma unroll
        for (int i = 0; i < NL; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) zb[(lq * 4 + r) * ZS + l15] = acc[i][r];
            __builtin_amdgcn_wave_barrier();
            const f32x4 row = *reinterpret_cast<const f32x4*>(zb + (lane >> 2) * ZS + (lane & 3) * 4);
            __builtin_amdgcn_wave_barrier();
            bstore16_sc1(rsEx, __builtin_bit_cast(u32x4v, row), (int)(((exc + (size_t)gate * nslots + (size_t)i * NC + pwave) * 256 + lane * 4) * 4), 0);
        }
    };

    int nl = live_tiles(T - 1);
    load_step(T - 1, nl);
    unsigned arrival = 0, carrival = 0;                        // grid-wide hand-offs so far; exchanges of this cluster so far
    for (int t = T - 1; t >= 0; --t) {
        float dh[PS];
#pragma unroll
        for (int i = 0; i < PS; ++i) dh[i] = 0.0f;
        if (t < T - 1) {
            gs.wait_all(arrival, pwave, lane);
            ++arrival;
            if (nl > 0) {                                         // (uniform over the cluster: the four gate workgroups of a part agree)
                bool done = false;
                static_for<1, TPP + 1>([&](auto k_) {
                    if (!done && nl == decltype(k_)::value) { product(k_, t); done = true; }
                });
                gs.arrive_one(ccount, tid);
                gs.wait_one(ccount, 4u * (carrival + 1u), pwave, lane);
                ++carrival;
#pragma unroll
                for (int i = 0; i < PS; ++i) {
                    float s_ = 0.0f;
                    if (i < nl) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            s_ += __uint_as_float(__hip_atomic_load((const gu32*)(g.ex + (exc + (size_t)q * nslots + slot_of(i)) * 256 + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    }
                    dh[i] = s_;
                }
            }
        }
        // ---- BasicLSTMCell backward pointwise (the expressions of lstm_bwd_pointwise_kernel), one (row, unit) per thread and slot
        float dzv[PS][4];
#pragma unroll
        for (int i = 0; i < PS; ++i) {
            float d = dx[i];
            if (g.keep < 1.0f && g.dext && t >= g.dext_t0)
                d = (d / g.keep) * dropout_keep01(g.seed_lo, g.seed_hi, vid[i], sid[i], g.drop_code0 + (uint32_t)t, (uint32_t)pu[i], g.keep);
            const float dht = dh[i] + d;
            const float si = sg[i][0], tj = sg[i][1], sf = sg[i][2], so = sg[i][3];
            const float tc = dm_tanhf(cnew[i]);
            const float dc = dht * so * (1.f - tc * tc) + dc_reg[i];
            dzv[i][0] = dc * tj * si * (1.f - si);
            dzv[i][1] = dc * si * (1.f - tj * tj);
            dzv[i][2] = dc * cprev[i] * sf * (1.f - sf);
            dzv[i][3] = dht * tc * so * (1.f - so);
            if (i < nl) dc_reg[i] = dc * sf;                     // (a tile that has not joined yet keeps dc = 0: its operands above are zeros)
        }
        if (t > 0) {
            // ---- dz_t -> the four gate images of the other parity, regrouped through LDS so that every thread writes ONE
            // 16-byte fragment slot: thread (gate q = tid / 64, slot L = tid % 64) takes row L % 16, units L / 16 + 4e
            const size_t inext = (size_t)(t & 1) * 4 * img_floats;
#pragma unroll
            for (int i = 0; i < PS; ++i) {
                if (!(i < nl)) continue;                         // (uniform over the workgroup)
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 4; ++q) dzl[(q * 16 + pr) * 17 + pn] = pok[i] ? dzv[i][q] : 0.0f;
                __syncthreads();
                const int q = tid >> 6, L = tid & 63, r = L & 15, kq = L >> 4;
                u32x4v w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = __float_as_uint(dzl[(q * 16 + r) * 17 + kq + 4 * e]);
                const size_t dst = inext + (size_t)q * img_floats + ((size_t)((tile0 + ptile[i] * TS) * NG + jj * NC + pc[i]) * 64 + L) * 4;
                bstore16_sc1(rsImg, w, (int)(dst * 4), 0);
            }
            gs.arrive(tid);
        }
        // ---- history: dZ[t] (read by the weight-gradient contractions after the launch), then the next step's operands
#pragma unroll
        for (int i = 0; i < PS; ++i) {
            if (!pok[i] || !(i < nl)) continue;
            float* zp = g.dZ + (size_t)t * g.dz_tstride + (size_t)pm[i] * 4 * H + pu[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) zp[(size_t)q * H] = dzv[i][q];
        }
        if (t > 0) {
            nl = live_tiles(t - 1);
            load_step(t - 1, nl);
        }
    }
}

// ---- backward recurrence: configurations, eligibility, launcher
typedef void (*BwdFn)(const BwdChainArgs);
struct BwdCfg { int ng, tmw, nc; BwdFn fn; const char* name; BwdFn fn_live; const char* name_live; };     // fn_live: the live-row variant of the register-weights form
const BwdCfg kBwd[] = {
    {8, 1, 1, lstm_bwd_chain_kernel<8, 1, 1>, "bchain(ng8,m64)"},     {8, 2, 1, lstm_bwd_chain_kernel<8, 2, 1>, "bchain(ng8,m128)"},
    {8, 4, 1, lstm_bwd_chain_kernel<8, 4, 1>, "bchain(ng8,m256)"},    {64, 1, 1, lstm_bwd_chain_kernel<64, 1, 1>, "bchain(ng64,m64)"},
    {64, 2, 1, lstm_bwd_chain_kernel<64, 2, 1>, "bchain(ng64,m128)"}, {64, 4, 1, lstm_bwd_chain_kernel<64, 4, 1>, "bchain(ng64,m256)"},
    // 32 units x half the row tiles per workgroup (tmw = row tiles per wave of a part): 257 .. 384 rows
    {64, 3, 2, lstm_bwd_chain_kernel<64, 3, 2>, "bchain2(ng64,m384)"},
    // weights in registers, 64 units x a gate x a quarter of the row tiles per workgroup (tmw = row tiles per part)
    {64, 5, 4, lstm_bwd_chain4_kernel<64, 5>, "bchain4(ng64,m320)", lstm_bwd_chain4_live_kernel<64, 5>, "bchain4(ng64,m320)[live]"},
    {64, 6, 4, lstm_bwd_chain4_kernel<64, 6>, "bchain4(ng64,m384)", lstm_bwd_chain4_live_kernel<64, 6>, "bchain4(ng64,m384)[live]"},
};
constexpr int kNumBwd = (int)(sizeof(kBwd) / sizeof(kBwd[0]));
int bwd_lds_bytes(const BwdCfg& c)
{
    if (c.nc == 4) return (3 * c.tmw * 8 * 256 + 4 * 16 * 20 + 4 * 16 * 17) * 4;      // the A ring (3 chunks of 8 groups x tmw tiles) instead of a W slice
    return (c.ng * c.nc * 256 + 4 * 16 * 20 + 4 * 16 * 17) * 4;
}
bool bwd_two_parts(int M, int H) { return M > 256 && (H + 15) / 16 > 8; }
bool bwd_four_parts(int M, int H)
{
    static const bool on = [] { const char* e = getenv("S2VT_BCHAIN4"); return !(e && e[0] == '0'); }();      // dev knob
    return on && bwd_two_parts(M, H);
}
int bwd_cfg(int M, int H)
{
    const int ng = (H + 15) / 16 <= 8 ? 8 : 64;
    if (bwd_four_parts(M, H)) {
        const int tpp = ((M + 15) / 16 + 3) / 4;
        for (int i = 0; i < kNumBwd; ++i)
            if (kBwd[i].nc == 4 && kBwd[i].ng == ng && kBwd[i].tmw == (tpp <= 5 ? 5 : 6)) return i;
        return -1;
    }
    const bool two = bwd_two_parts(M, H);
    const int tmw = two ? 3 : (M <= 64 ? 1 : (M <= 128 ? 2 : 4)), nc = two ? 2 : 1;
    for (int i = 0; i < kNumBwd; ++i)
        if (kBwd[i].ng == ng && kBwd[i].tmw == tmw && kBwd[i].nc == nc) return i;
    return -1;
}
struct BwdDev { std::once_flag once; bool ok = false; int per_cu[kNumBwd] = {}; };
constexpr int kMaxDev = 32;
BwdDev g_bdev[kMaxDev];
BwdDev* bwd_dev_state()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
    BwdDev& d = g_bdev[dev];
    std::call_once(d.once, [&d] {
        bool ok = true;
        for (int i = 0; ok && i < kNumBwd; ++i) {
            const BwdCfg& c = kBwd[i];
            ok = hipFuncSetAttribute(reinterpret_cast<const void*>(c.fn), hipFuncAttributeMaxDynamicSharedMemorySize, bwd_lds_bytes(c)) == hipSuccess;
            if (ok && c.fn_live)
                ok = hipFuncSetAttribute(reinterpret_cast<const void*>(c.fn_live), hipFuncAttributeMaxDynamicSharedMemorySize, bwd_lds_bytes(c)) == hipSuccess;
            int n = 0;
            if (ok && hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(c.fn), 256, bwd_lds_bytes(c)) == hipSuccess)
                d.per_cu[i] = n;
        }
        d.ok = ok;
    });
    return &d;
}

}  // namespace

constexpr int kBwdMaxRows = 384;

bool bwd_chain_eligible(int M, int H)
{
    static const bool off = [] { const char* e = getenv("S2VT_BCHAIN"); return e && e[0] == '0'; }();      // dev knob: per-step launches
    if (off || chain_persistent_disabled()) return false;
    ChainHost hst;
    BwdDev* b = bwd_dev_state();
    if (!chain_host(&hst) || !b || !b->ok) return false;
    const ChainHost* d = &hst;
    if (!(M >= 1 && M <= kBwdMaxRows && H >= 4 && (H & 3) == 0 && H <= 1024)) return false;
    const int ci = bwd_cfg(M, H);
    if (ci < 0) return false;
    const int nc = kBwd[ci].nc, ncg = (H + 16 * nc - 1) / (16 * nc);
    return (long)b->per_cu[ci] * d->num_cus >= 4L * nc * ncg;   // every ACTIVE workgroup fits on the chip at once
}

bool bwd_chain_live_capable(int M, int H)
{
    if (!bwd_chain_auto(M, H)) return false;
    const int ci = bwd_cfg(M, H);
    return ci >= 0 && kBwd[ci].fn_live != nullptr;
}

bool bwd_chain_auto(int M, int H)
{
    static const int maxm = [] { const char* e = getenv("S2VT_BCHAIN_MAXM"); return e ? atoi(e) : 256; }();  // rows up to which the one-part form is chosen unasked (M = 256: 1.01 ms against 1.39 as launches)
    static const bool two = [] { const char* e = getenv("S2VT_BCHAIN2"); return !(e && e[0] == '0'); }();      // the two-part form (257 .. 384 rows)
    return (M <= maxm || (two && bwd_two_parts(M, H))) && bwd_chain_eligible(M, H);
}

// geometry of a launch: column tiles per workgroup, row tiles per part, unit groups
static void bwd_geometry(int M, int H, int* nc, int* tmw, int* tpp, int* ncg)
{
    const bool two = bwd_two_parts(M, H);
    const int tiles = (M + 15) / 16;
    if (bwd_four_parts(M, H)) {
        *nc = 4;
        *tpp = (tiles + 3) / 4 <= 5 ? 5 : 6;
        *tmw = *tpp;
        *ncg = (H + 63) / 64;
        return;
    }
    *nc = two ? 2 : 1;
    *tmw = two ? 3 : (M <= 64 ? 1 : (M <= 128 ? 2 : 4));
    *tpp = two ? (tiles + 1) / 2 : 4 * *tmw;
    *ncg = (H + 16 * *nc - 1) / (16 * *nc);
}

void bwd_chain_scratch(int H, int M, size_t* img_floats, size_t* ex_floats, size_t* sync_bytes)
{
    const int ng = (H + 15) / 16 <= 8 ? 8 : 64;
    int nc, tmw, tpp, ncg;
    bwd_geometry(M, H, &nc, &tmw, &tpp, &ncg);
    *img_floats = (size_t)8 * nc * tpp * ng * 256;                             // 2 parities x 4 gates x (parts * tpp) row tiles
    *ex_floats = (size_t)ncg * nc * 4 * tpp * nc * 256;                        // clusters x 4 gates x (row tile, column tile) slots
    *sync_bytes = kChainSyncBytes + (size_t)ncg * nc * 128;
}

hipError_t launch_lstm_bwd_chain(const BwdChainLaunch& a, hipStream_t st)
{
    if (!bwd_chain_eligible(a.M, a.H)) return hipErrorInvalidValue;
    if (a.T <= 0) return hipSuccess;
    if ((reinterpret_cast<uintptr_t>(a.W) & 15) || (a.ldw & 3) || (reinterpret_cast<uintptr_t>(a.img) & 15) || (reinterpret_cast<uintptr_t>(a.ex) & 15))
        return hipErrorInvalidValue;
    ChainHost hst;
    if (!chain_host(&hst)) return hipErrorInvalidValue;
    const int dev = hst.device;
    const int ci = bwd_cfg(a.M, a.H);
    const BwdCfg& c = kBwd[ci];
    BwdChainArgs k;
    std::memset(&k, 0, sizeof(k));
    k.W = a.W; k.ldw = a.ldw; k.kw0 = a.kw0;
    k.gates = a.gates; k.gates_tstride = a.gates_tstride; k.C = a.C; k.state_tstride = a.state_tstride;
    k.dext = a.dext; k.dext_tstride = a.dext_tstride; k.ld_ext = a.ld_ext; k.dext_t0 = a.dext_t0;
    k.dZ = a.dZ; k.dz_tstride = a.dz_tstride; k.M = a.M; k.H = a.H; k.T = a.T;
    k.keep = a.keep; k.seed_lo = a.seed_lo; k.seed_hi = a.seed_hi; k.drop_code0 = a.drop_code0;
    k.video_id = a.video_id; k.sample_id = a.sample_id;
    k.img = a.img; k.ex = a.ex; k.sync = a.sync;
    const bool live = a.perm && a.nlive && c.fn_live;
    if (live) { k.perm = a.perm; k.nlive = a.nlive; }
    k.status = hst.status_dev; k.fault = hst.fault; k.spin_limit = hst.spin_limit;
    int nc_, tmw_, tpp_, ncg_;
    bwd_geometry(a.M, a.H, &nc_, &tmw_, &tpp_, &ncg_);
    k.ncg = ncg_; k.tpp = tpp_; k.img_tiles = nc_ * tpp_;
    size_t imgf, exf, syncb;
    bwd_chain_scratch(a.H, a.M, &imgf, &exf, &syncb);
    ChainLaunchOrder order;                                    // one persistent grid at a time per process
    {
        hipError_t we = order.before(st, dev);
        if (we != hipSuccess) return we;
    }
    ZeroList z;
    z.add(a.sync, syncb); z.add(a.img, imgf * 4);              // (rows >= M and k >= H of the images must read as zeros)
    hipError_t e = launch_zero_regions(z, st);
    if (e != hipSuccess) return e;
    e = chain_gate_zeroed(st);                                 // (gated overlap: a side stream may poll the counters from here on)
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)((k.ncg + 7) / 8 * 32 * nc_));
    const double flops = 2.0 * a.M * (double)a.H * 4.0 * a.H * (a.T - 1);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const int pci = live ? ci + 100 : ci;                      // (the live-row variant is a profiler row of its own)
    const bool prof = prof_wants(6, pci);
    if (prof) {
        hipError_t pe = prof_events(&e0, &e1);
        if (pe != hipSuccess) return pe;
        (void)hipEventRecord(e0, st);
    }
    hipLaunchKernelGGL(live ? c.fn_live : c.fn, grid, dim3(256), bwd_lds_bytes(c), st, k);
    if (prof) {
        (void)hipEventRecord(e1, st);
        prof_record(6, pci, live ? c.name_live : c.name, live ? 0.0 : flops, e0, e1);     // (live rows: the executed count lives on the device -- no rate is claimed)
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = chain_gate_launched(a.sync, 4u * (unsigned)nc_ * (unsigned)k.ncg);     // every active workgroup arrives once per iteration
    if (e != hipSuccess) return e;
    return order.after(st, dev);
}

}  // namespace s2vt
