// chain.hip -- a whole BasicLSTMCell recurrence (T steps, M <= 64 rows) in ONE launch: the unroll of tf_s2vt.py:113-153
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
// All workgroups must be co-resident (grid = H / 4 <= CU count, one 256-thread workgroup per CU); every spin is bounded
// and a timeout sets a status word instead of hanging (s2vt_chain_status).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>

#include "detmath.h"
#include "internal.h"

namespace s2vt {

namespace {

typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ void bload16_sc1(f32x4& d, uint32_t voff, i32x4 rsrc, uint32_t soff)
{
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen sc1" : "=v"(d) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

constexpr int kShards = 8;             // counter shards, one 128-byte line each; word kShards * 32 = timeout flag
constexpr unsigned kSpinLimit = 1u << 21;

template <int NG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_chain_kernel(const ChainArgs g)
{
    constexpr int ZS = 20;                                     // z tile row stride (floats): 16-byte aligned rows, conflict-light
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                                          // [NG][64 lanes][4]: B fragments of this workgroup's 16 gate columns
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* zb = smem + NG * 256 + wave * (16 * ZS);
    const int l15 = lane & 15, lq = lane >> 4;
    const int H = g.H, M = g.M, T = g.T;
    const int u0 = blockIdx.x * 4;
    const int nwg = gridDim.x;

    // ---- this workgroup's slice of the recurrent rows -> LDS, once.  Column c = uu * 4 + gate of the slice is W column
    // gate * H + u0 + uu; element (k, c) goes to group k / 16, lane (k % 4) * 16 + c, component (k % 16) / 4.
    for (int idx = tid; idx < NG * 16 * 4; idx += 256) {
        const int k = idx >> 2, gt = idx & 3;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < H) v = *reinterpret_cast<const f32x4*>(g.W + (size_t)(g.kw0 + k) * g.ldw + (size_t)gt * H + u0);
        const int j = k >> 4, e = (k & 15) >> 2, kq = k & 3;
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) Wl[((j * 64 + kq * 16 + uu * 4 + gt) << 2) + e] = v[uu];
    }

    // ---- the (row, unit) this lane finishes at every step
    const int rt = lane >> 2, uu = lane & 3;                   // row within the wave's 16-row tile, unit within the 4
    const int row = wave * 16 + rt, u = u0 + uu;
    const bool rok = row < M;
    const float bi = g.bias[u], bj = g.bias[H + u], bf = g.bias[2 * H + u], bo = g.bias[3 * H + u];
    float c_reg = (rok && g.c0) ? g.c0[(size_t)row * H + u] : 0.0f;
    uint32_t vid = 0, sid = 0;
    if (g.keep < 1.0f && rok) { vid = (uint32_t)g.video_id[row]; sid = (uint32_t)g.sample_id[row]; }
    // where this lane's h value sits in the A-fragment image: k = u -> group u / 16, lane (u % 4) * 16 + rt, component (u % 16) / 4
    const size_t a_own = ((size_t)(wave * NG + (u >> 4)) * 64 + (size_t)((u & 3) * 16 + rt)) * 4 + ((u & 15) >> 2);
    float* const abuf0 = g.abuf;
    float* const abuf1 = g.abuf + (size_t)4 * NG * 256;

    gu32* const sync = (gu32*)g.sync;
    auto arrive = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // EVERY storing wave drains its write-through stores ...
        __syncthreads();                                       // ... before the ONE lane that signals for all of them
        if (tid == 0) __hip_atomic_fetch_add(sync + (blockIdx.x & (kShards - 1)) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    bool dead = false;                                         // a wait timed out: stop waiting, finish the launch
    auto wait_all = [&](unsigned arrival) __attribute__((always_inline)) {     // every workgroup has made arrival number `arrival`
        if (wave == 0 && !dead) {
            const unsigned mine = lane < kShards ? (unsigned)((nwg + kShards - 1 - lane) / kShards) * (arrival + 1u) : 0u;
            unsigned spins = 0;
            for (;;) {
                const unsigned v = lane < kShards ? __hip_atomic_load(sync + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                if (__all(lane >= kShards || v >= mine)) break;
                if (++spins > kSpinLimit) {                    // never hang: flag it and go on (results are then garbage, status says so)
                    if (lane == 0) {
                        __hip_atomic_store(sync + kShards * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (g.status) __hip_atomic_fetch_add(g.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // host-visible
                    }
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
    };

    // ---- arrival 0: h_0 in fragment order (the image is zero-filled by the launcher: rows >= M and k >= H stay zero)
    if (rok) {
        const float h0 = g.h0 ? g.h0[(size_t)row * H + u] : 0.0f;
        __hip_atomic_store((gu32*)(abuf0 + a_own), __float_as_uint(h0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the carried partial of step 0 (accumulator layout: lane (column l15, row group lq) holds rows lq*4 + r)
    const int ccol = (l15 & 3) * H + u0 + (l15 >> 2);          // W / cinit column of slice column l15
    float ci[4] = {0.f, 0.f, 0.f, 0.f};
    auto load_cinit = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = wave * 16 + lq * 4 + r;
            ci[r] = (g.cinit && t < g.cinit_steps && m < M) ? g.cinit[(size_t)t * g.cinit_tstride + (size_t)m * g.ldcinit + ccol] : 0.0f;
        }
    };
    load_cinit(0);
    arrive();

    for (int t = 0; t < T; ++t) {
        f32x4 acc = {ci[0], ci[1], ci[2], ci[3]};
        asm volatile("" : "+v"(acc));                          // the partial is in registers before the ring below is issued
        wait_all((unsigned)t);
        // ---- all A fragments of this step in flight at once (asm-issued: hipcc neither sinks nor counts them)
        const float* acur = (t & 1) ? abuf1 : abuf0;
        const i32x4 rsA = make_rsrc(acur + (size_t)wave * NG * 256);
        const uint32_t voff = (uint32_t)lane * 16u;
        // Ring of RING groups: group j + RING is issued into group j's registers as soon as its MFMAs have read them, so RING
        // groups (32 KB per wave) stay in flight through the first NG - RING groups and the counted wait is a constant.
        // (The whole step at once -- 256 registers -- made hipcc move ring registers to AGPRs right behind the asm load,
        // i.e. before the data had landed.)
        constexpr int RING = NG < 32 ? NG : 32;
        f32x4 a[RING];
        static_for<0, RING>([&](auto j_) { constexpr int j = decltype(j_)::value; bload16_sc1(a[j], voff, rsA, (uint32_t)j * 1024u); });
        const f32x4* bl = reinterpret_cast<const f32x4*>(Wl) + lane;
        constexpr int PB = NG < 4 ? NG : 4;                    // B fragments read PB groups ahead
        f32x4 b[PB];
        static_for<0, PB>([&](auto j_) { constexpr int j = decltype(j_)::value; b[j] = bl[j * 64]; });
        static_for<0, NG>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            // younger loads still wanted in flight when group j is needed: RING - 1 while refills keep pace, then the tail
            wait_vmcnt<(j + RING <= NG ? RING - 1 : NG - 1 - j)>();
            pin(a[j % RING]);
            const f32x4 bj4 = b[j % PB];
            if constexpr (j + PB < NG) b[j % PB] = bl[(j + PB) * 64];
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j % RING][0], bj4[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j % RING][1], bj4[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j % RING][2], bj4[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j % RING][3], bj4[3], acc, 0, 0, 0);
            if constexpr (j + RING < NG) {
                __builtin_amdgcn_sched_barrier(0);             // the refill stays BEHIND the MFMAs that read these registers
                bload16_sc1(a[j % RING], voff, rsA, (uint32_t)(j + RING) * 1024u);
            }
        });
        // ---- gates of a unit meet through the wave's LDS tile: z[row][uu*4 + gate]
#pragma unroll
        for (int r = 0; r < 4; ++r) zb[(lq * 4 + r) * ZS + l15] = acc[r];
        __builtin_amdgcn_wave_barrier();
        const f32x4 z = *reinterpret_cast<const f32x4*>(zb + rt * ZS + uu * 4);
        __builtin_amdgcn_wave_barrier();
        // BasicLSTMCell pointwise (gate order i, j, f, o; forget_bias 1.0 added at run time) -- the EPI_LSTM expressions
        const float zi = z[0] + bi, zj = z[1] + bj, zf = z[2] + bf, zo = z[3] + bo;
        const float si = dm_sigmoidf(zi);
        const float tj = dm_tanhf(zj);
        const float sf = dm_sigmoidf(zf + 1.0f);
        const float so = dm_sigmoidf(zo);
        const float t1 = c_reg * sf;
        const float t2 = si * tj;
        const float c = t1 + t2;
        const float h = dm_tanhf(c) * so;
        c_reg = c;
        // The hand-off first: h_t write-through, drained and signalled BEFORE the history stores, which nobody in this
        // launch reads -- they complete under the other workgroups' arrival (and only have to by the end of the kernel).
        if (t + 1 < T) {
            if (rok)
                __hip_atomic_store((gu32*)(((t & 1) ? abuf0 : abuf1) + a_own), __float_as_uint(h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            arrive();
        }
        if (rok) {
            const size_t o = (size_t)row * H + u;
            g.C[(size_t)(t + 1) * g.state_tstride + o] = c;
            g.Hh[(size_t)(t + 1) * g.state_tstride + o] = h;
            if (g.out) {
                float ov = h;
                if (g.keep < 1.0f) ov = (h / g.keep) * dropout_keep01(g.seed_lo, g.seed_hi, vid, sid, g.drop_code0 + (uint32_t)t, (uint32_t)u, g.keep);
                g.out[(size_t)t * g.out_tstride + o] = ov;
            }
            if (g.gates) {
                float* gp = g.gates + (size_t)t * g.gates_tstride + (size_t)row * 4 * H + u;
                gp[0] = si; gp[H] = tj; gp[2 * H] = sf; gp[3 * H] = so;
            }
        }
        if (t + 1 < T) load_cinit(t + 1);
    }
}

typedef void (*ChainFn)(const ChainArgs);
struct ChainCfg { int ng; ChainFn fn; };
const ChainCfg kChain[] = {{8, lstm_chain_kernel<8>}, {64, lstm_chain_kernel<64>}};

std::once_flag g_chain_once;
int g_num_cus = 0;
unsigned* g_status_host = nullptr;     // pinned, device-mapped: timeouts of every chain launch of this process
unsigned* g_status_dev = nullptr;

}  // namespace

size_t chain_scratch_floats(int H)
{
    const int ng = (H + 15) / 16 <= 8 ? 8 : 64;
    return (size_t)2 * 4 * ng * 256;                          // two fragment images of h (ping-pong)
}

bool chain_eligible(int M, int H)
{
    static const bool off = [] { const char* e = getenv("S2VT_CHAIN"); return e && e[0] == '0'; }();
    std::call_once(g_chain_once, [] {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) g_num_cus = p.multiProcessorCount;
        bool ok = g_num_cus > 0;
        for (const ChainCfg& c : kChain)
            ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(c.fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (c.ng * 256 + 4 * 16 * 20) * 4) == hipSuccess;
        void* hp = nullptr;
        if (ok && hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess) {
            g_status_host = static_cast<unsigned*>(hp);
            *g_status_host = 0u;
            void* dp = nullptr;
            if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) g_status_dev = static_cast<unsigned*>(dp);
        }
        if (!ok || !g_status_dev) g_num_cus = 0;               // an LDS request refused / no status word: the per-step path serves
    });
    return !off && M >= 1 && M <= 64 && H >= 4 && (H & 3) == 0 && H <= 1024 && H / 4 <= g_num_cus;
}

hipError_t launch_lstm_chain(const ChainArgs& a, hipStream_t st)
{
    if (!chain_eligible(a.M, a.H)) return hipErrorInvalidValue;
    if (a.T <= 0) return hipSuccess;
    if ((reinterpret_cast<uintptr_t>(a.W) & 15) || (a.ldw & 3) || (reinterpret_cast<uintptr_t>(a.abuf) & 15)) return hipErrorInvalidValue;
    const ChainCfg& c = (a.H + 15) / 16 <= 8 ? kChain[0] : kChain[1];
    ChainArgs a2 = a;
    a2.status = g_status_dev;
    // every polled word and the fragment images start from zero on EVERY call (a memset node ahead of the launch)
    hipError_t e = hipMemsetAsync(a.sync, 0, kChainSyncBytes, st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(a.abuf, 0, chain_scratch_floats(a.H) * 4, st);
    if (e != hipSuccess) return e;
    const int lds = (c.ng * 256 + 4 * 16 * 20) * 4;
    const dim3 grid((unsigned)(a.H / 4));
    const double flops = 2.0 * a.M * (double)a.H * 4.0 * a.H * a.T;
    if (!prof_wants(5, c.ng == 8 ? 0 : 1)) {
        hipLaunchKernelGGL(c.fn, grid, dim3(256), lds, st, a2);
        return hipGetLastError();
    }
    hipEvent_t e0, e1;
    hipError_t pe = prof_events(&e0, &e1);
    if (pe != hipSuccess) return pe;
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL(c.fn, grid, dim3(256), lds, st, a2);
    (void)hipEventRecord(e1, st);
    prof_record(5, c.ng == 8 ? 0 : 1, c.ng == 8 ? "chain(ng8)" : "chain(ng64)", flops, e0, e1);
    return hipGetLastError();
}

unsigned chain_timeouts() { return g_status_host ? *static_cast<volatile unsigned*>(g_status_host) : 0u; }

}  // namespace s2vt
