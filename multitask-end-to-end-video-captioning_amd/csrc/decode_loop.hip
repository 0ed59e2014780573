// decode_loop.hip -- the sampler's whole decode loop (reinforcement_multisampling_tf_s2vt.py:318-337: Tc steps of
// {embedding lookup of the previous pick, LSTM2, vocabulary logits, multinomial / argmax pick} for the R = (K + 1) B rows
// that advance together) as ONE persistent launch, instead of 2 Tc launches {LSTM2 step, vocabulary pick}.
//
//   256 workgroups, one per CU, EIGHT waves (two per SIMD).  A step is two phases separated by grid-wide hand-offs
//   (chain_common.h GridSync, the forms of the persistent recurrences):
//
//   A  LSTM2 step (waves 0-3; waves 4-7 only keep the workgroup barriers: with four waves the phase already runs at the
//      pipe's dependent-issue rate, and its 250 registers per wave are what two waves per SIMD leave) -- the inner loop of
//      decode4.hip: workgroup (unit group cg of 16 hidden units, row part rp) multiplies its
//      TPP row tiles with the fragment-order W2' stream (global -> registers) and the A fragments of the embedding rows of the
//      previous picks (Wemb', gathered by the DMA source offset) and of h2_{t-1} (the state IMAGE [tile][k group][64][4],
//      write-through stores / sc1 loads: what chain.hip exchanges), continues the chain from the carried partial P2_t,
//      BasicLSTMCell pointwise with c_t kept in registers for the whole loop, h_t -> the other image.            [arrive / wait]
//   B  vocabulary pick (all eight waves: at ONE wave per SIMD nothing hid the latencies of its loop and of the pick epilogue --
//      84 + 30 us against 70 + 11 in the launch form, profiles/r03_decode_loop_probe.jsonl) -- the layout of
//      tools/micro/pick_phase.hip: workgroup b owns 48 vocabulary columns for ALL rows; every wave streams the image blocks of
//      its own three row tiles straight into registers (sc1), the 48 columns of embed_word_W go
//      global -> LDS by buffer_load ... lds in 16-row stages; epilogue = the PICK epilogue of gemm_mfma.h (bias, two-tier
//      Gumbel-max on the Philox stream of (video, sample, step, column), per-row maximum over the 16 lanes of a row) and one
//      agent-scope 64-bit atomic max per row into the packed pick words, which phase A of the next step reads.   [arrive / wait]
//
// Same arithmetic as the launches it replaces: each z and each logit is the same ascending-k fmaf chain (carried partial,
// embedding rows, recurrent rows; then + bias), the keys are the same expressions, ties go to the lowest column -- ids are
// bit-identical (tests/test_gpu_decode_loop.py runs the sampler both ways in child processes: four shapes, two seeds each).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "chain_common.h"

namespace s2vt {

namespace {

struct DecLoopArgs {
    // phase A
    const float* wemb_p; int erow;                 // Wemb' [V][erow]
    const float* w2_p;                             // W2' [ncg][4][ngt][256]
    const float* bias2;                            // [4H]
    const float* P2; size_t p2_tstride; int ldp2; int B;     // carried partial of decode step t: P2 + t * p2_tstride, row % B
    const float* c0;                               // c2 at the start of the decoding stage [B, H] (row % B)
    float* himg0; float* himg1;                    // state images; himg0 holds h2 at the start
    unsigned long long* packed; int pick_stride;   // [Tc][R][pick_stride] packed picks, zeroed before the launch
    // phase B
    const float* Wout; int ldwo; const float* bout;
    int video_base;                                // noise-stream ids as sampler_rows_kernel assigns them: video = video_base + row % B, sample = row / B
    uint32_t seed_lo, seed_hi;
    int noise_rows;                                // rows [0, noise_rows) draw Gumbel noise (sample_id >= 0), the rest are argmax rows
    int R, H, V, Tc;
    int eg, hg, hgp, ncg, ech, hch;
    unsigned* sync; unsigned* status; unsigned* fault; unsigned spin_limit;
};

constexpr int kCG = 8, kNBUF = 3;                  // phase A: k-groups per chunk, LDS chunk buffers (decode4.hip's values)
constexpr int kTNC = 3;                            // phase B: column tiles (16 vocabulary columns each) per workgroup
// phase B at R <= 64 (round 6): waves 4-7 are LOADER waves -- state image and embed_word_W both go global -> LDS by DMA in stages of kKG k-groups,
// kNSTG stages deep, and waves 0-3 only read fragments and multiply (tools/micro/stream_probe.hip: a wave that issues a vector-memory instruction is
// held ~170 clocks, and this phase spent more wave-time issuing its own loads than multiplying: 28 us for 10 us of MFMAs)
#ifndef S2VT_DL_KG
#define S2VT_DL_KG 4
#endif
#ifndef S2VT_DL_NSTG
#define S2VT_DL_NSTG 4
#endif
constexpr int kKG = S2VT_DL_KG, kNSTG = S2VT_DL_NSTG;
constexpr int kB1Base = 8192;                      // floats: phase B's stages start behind phase A's ring + gate tiles (7424 floats at TPP = 1)
constexpr int kB1Stage = 4 * kKG * 256 + kKG * 16 * 48;    // floats per stage: 4 row tiles x kKG image blocks + kKG x 16 rows x 48 columns of W

#ifdef S2VT_DL_STAMP
__device__ unsigned long long dl_stamp_acc[16];
#define DL_STAMP(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); if (blockIdx.x == 40 && threadIdx.x == 0) atomicAdd(&dl_stamp_acc[i], n_ - st_prev); st_prev = n_; } while (0)
// (loader wave 4 of the same workgroup: slots 8.. of the same table)
#define DL_STAMP_L(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); if (blockIdx.x == 40 && threadIdx.x == 256) atomicAdd(&dl_stamp_acc[i], n_ - stl_prev); stl_prev = n_; } while (0)
#else
#define DL_STAMP(i) do { } while (0)
#define DL_STAMP_L(i) do { } while (0)
#endif

template <int TPP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void decode_loop_kernel(const DecLoopArgs g)
{
#ifdef S2VT_DL_STAMP
    unsigned long long st_prev = __builtin_readcyclecounter();
    unsigned long long stl_prev = st_prev;
#endif
    constexpr int ZS = 20;
    constexpr int CG = kCG, NBUF = kNBUF;
    constexpr int PPG = TPP;
    constexpr int CHF = CG * PPG * 256;                        // floats per chunk buffer
    constexpr int GPW = CG / 4;
    constexpr int YOUNGER = (NBUF - 2) * GPW * TPP + (NBUF - 1) * CG;
    static_assert(YOUNGER <= 63, "vmcnt range");
    // (R <= 64: a k-group is 12 MFMAs per wave instead of 108 -- the W stages and the A fragments must be requested 3x as many groups
    //  ahead to cover the same latency: with the 384-row depths the loop ran 1070 cycles per group against 384 of MFMAs, round-5 stamps)
    constexpr int TNC = kTNC;
    constexpr int ZA0 = TPP == 1 ? kB1Base + kNSTG * kB1Stage : 0;   // floats: the accumulator slots of the pick epilogue -- behind the stages (R <= 64), or IN the ring once it is dead (257-384 rows)
    constexpr int TMW = TPP;                                   // phase B: row tiles per MFMA wave (waves 0-3: tile w * TPP + i of the image's 4 TPP); waves 4-7 load

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ab = smem;                                          // phase A: [NBUF][CG][PPG][64][4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* zb = smem + NBUF * CHF + (wave & 3) * (16 * ZS);
    const int l15 = lane & 15, lq = lane >> 4;
    const int H = g.H, R = g.R;
    typedef __attribute__((address_space(3))) void* lds_ptr;

    // ---- phase A identity: the four row parts of a unit group share blockIdx % 8 (one XCD: one fetch of the W2' slice)
    const int cg = ((int)blockIdx.x >> 5) * 8 + ((int)blockIdx.x & 7), rp = ((int)blockIdx.x >> 3) & 3;
    const bool aact = cg < g.ncg;                              // (256 workgroups, 4 * ncg <= 252 of them own LSTM2 work)
    const int u0 = cg * 16 + wave * 4;
    const bool wact = aact && u0 < H;
    const int tb = rp * TPP;
    const int ech = g.ech, hch = g.hch, nch = ech + hch;
    const int ngt = nch * CG;
    // ---- phase B identity: 48 vocabulary columns
    const int nct = (g.V + 15) / 16;
    const int n0 = (int)blockIdx.x * (16 * TNC);
    const bool bact = (int)blockIdx.x * TNC < nct;
    const int kg = g.hg;

    GridSync gs{(gu32*)g.sync, g.status, g.fault, g.spin_limit, (int)gridDim.x, false};

    // ---- loop-invariant operands
    const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.wemb_p), 0, (int)0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.w2_p + ((size_t)(aact ? cg : 0) * 4 + (wave & 3)) * ngt * 256), 0, ngt * 1024, 0x00020000);
    const int ccol = (l15 & 3) * H + u0 + (l15 >> 2);
    // carried partial P2_t[row % B]: B is a multiple of 16, so row % B = (tile's first row) % B + row in tile -- one uniform
    // remainder per row tile, kept in scalar registers; the byte offsets are rebuilt every step (two VALU per element)
    int tmod[TPP];
#pragma unroll
    for (int i = 0; i < TPP; ++i) tmod[i] = __builtin_amdgcn_readfirstlane(((tb + i) * 16) % g.B);
    auto load_cinit = [&](int t, float (&ci)[TPP][4]) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.P2 + (size_t)t * g.p2_tstride), 0, (int)0x80000000u, 0x00020000);
#pragma unroll
        for (int i = 0; i < TPP; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = (tb + i) * 16 + lq * 4 + r;
                const int off = (wact && m < R) ? ((tmod[i] + lq * 4 + r) * g.ldp2 + ccol) * 4 : (int)0x80000000u;
                ci[i][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsC, off, 0, 0));
            }
    };
    // c_{t-1} of the (row, unit) pairs this lane finishes: in registers for the whole loop
    const int rt = lane >> 2, uu = lane & 3;
    const int u = u0 + uu;
    float c_reg[TPP];
#pragma unroll
    for (int i = 0; i < TPP; ++i) {
        const int row = (tb + i) * 16 + rt;
        c_reg[i] = (wact && row < R) ? g.c0[(size_t)(tmod[i] + rt) * H + u] : 0.0f;
    }
    float bi = 0.f, bj = 0.f, bf = 0.f, bo = 0.f;
    if (wact) { bi = g.bias2[u]; bj = g.bias2[H + u]; bf = g.bias2[2 * H + u]; bo = g.bias2[3 * H + u]; }
    // phase B epilogue operands: bias of this lane's columns, noise ids of its rows
    float ep_bias[TNC];
#pragma unroll
    for (int j = 0; j < TNC; ++j) {
        const int col = n0 + j * 16 + l15;
        ep_bias[j] = (bact && col < g.V) ? g.bout[col] : 0.0f;
    }
    DL_STAMP(0);

    for (int t = 0; t < g.Tc; ++t) {
        const float* him_in = (t & 1) ? g.himg1 : g.himg0;
        float* him_out = (t & 1) ? g.himg0 : g.himg1;
        // R <= 64: the loader waves (4-7) fetch embed_word_W for phase B; stage `sidx` of kKG x 16 rows x 48 columns goes to slot `slot` of
        // phase B's LDS ring (behind phase A's region).  The first kNSTG - 1 stages are issued HERE, before phase A: they depend on nothing.
        auto issue_w = [&](int sidx, int slot) __attribute__((always_inline)) {
            if constexpr (TPP == 1) {
                constexpr int KGc = kKG, WPWc = 12 * kKG / 16;
                const int lw = wave - 4;
                const int nstw = (g.hg + KGc - 1) / KGc;
                float* const sb = smem + kB1Base + slot * kB1Stage + 4 * KGc * 256;
                const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(g.Wout + (size_t)(sidx < nstw ? sidx : 0) * KGc * 16 * g.ldwo), 0, 0x7fffffff, 0x00020000);
                static_for<0, WPWc>([&](auto n_) {
                    constexpr int n = decltype(n_)::value;
                    const int q = (lw + 4 * n) * 64 + lane, brow = q / 12, bc4 = (q % 12) * 4;
                    const int k = sidx * KGc * 16 + brow;
                    const unsigned off = (sidx < nstw && k < H && n0 + bc4 < g.V) ? (unsigned)(brow * g.ldwo + n0 + bc4) * 4u : 0x80000000u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(sb + (lw + 4 * n) * 256), 16, off, 0, 0, 0);
                });
            }
        };
        if constexpr (TPP == 1) {
            if (bact && wave >= 4) static_for<0, kNSTG - 1>([&](auto j_) { issue_w(decltype(j_)::value, decltype(j_)::value); });
        }
        // =============================================================== phase A: LSTM2 step t
        if (aact && wave >= 4) {
            // waves 4-7 have no part in phase A: they keep its workgroup barriers (one per chunk, two behind the loop)
            // (bare s_barrier: these waves touch nothing of phase A, and at R <= 64 their phase-B weight prefetch must stay in flight --
            //  __syncthreads() would drain it in front of the first barrier and hold waves 0-3 there)
            for (int c0 = 0; c0 < nch; c0 += NBUF)
#pragma unroll
                for (int k = 0; k < NBUF; ++k) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
        } else if (aact) {
            // tokens of this wave's DMA rows: the previous step's picks (agent-scope loads: the atomics' home), <bos> = 1 at t = 0
            int tokoff[TPP];
#pragma unroll
            for (int i = 0; i < TPP; ++i) {
                const int m = (tb + i) * 16 + l15;
                int tk = 1;
                if (t > 0 && m < R) {
                    const unsigned long long wd = __hip_atomic_load(g.packed + ((size_t)(t - 1) * R + m) * g.pick_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    tk = (int)(~(uint32_t)wd);
                }
                if (m >= R || tk < 0 || tk >= g.V) tk = 0;
                tokoff[i] = tk * g.erow * 4 + lq * 16;
            }
            const __amdgpu_buffer_rsrc_t rsH =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(him_in + (size_t)tb * g.hgp * 256), 0, TPP * g.hgp * 1024, 0x00020000);
            constexpr int NPA = GPW * TPP, NP = NPA + CG;
            auto issue_piece = [&](auto j_, int c, float* dstb, f32x4* bdst) __attribute__((always_inline)) {
                constexpr int j = decltype(j_)::value;
                const int oob = c >= nch ? (int)0x80000000u : 0;
                if constexpr (j < NPA) {
                    constexpr int gl = j / TPP, i = j % TPP;
                    const int gq = wave + 4 * gl;
                    const int Ga = c * CG + gq;
                    const bool emb = c < ech;
                    const int Gh = Ga - ech * CG;
                    const bool inr = emb ? Ga < g.eg : Gh < g.hgp;
                    const int vo = (inr ? (emb ? tokoff[i] + Ga * 64 : lane * 16) : (int)0x80000000u) | oob;
                    const int so = (emb || !inr) ? 0 : (Gh + i * g.hgp) * 1024;
                    // aux 16 = sc1: the image is this launch's hand-off (the embedding rows take the same form: the two sources
                    // differ in operands only, not in control flow)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(emb ? rsE : rsH, (lds_ptr)(dstb + (gq * PPG + i) * 256), 16, vo, so, 0, 16);
                } else {
                    constexpr int gq = j - NPA;
                    const int sb = (oob ? 0 : c * CG) * 1024;
                    bdst[gq] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (lane * 16) | oob, sb + gq * 1024, 0));
                }
            };
            auto issue_chunk = [&](int c, float* dstb, f32x4* bdst) __attribute__((always_inline)) {
                static_for<0, NP>([&](auto j_) { issue_piece(j_, c, dstb, bdst); });
            };
            f32x4 acc[TPP];
            {
                float ci[TPP][4];
                load_cinit(t, ci);                                  // (its latency passes with the token loads')
#pragma unroll
                for (int i = 0; i < TPP; ++i) {
                    acc[i] = f32x4{ci[i][0], ci[i][1], ci[i][2], ci[i][3]};
                    asm volatile("" : "+v"(acc[i]));
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // tokens and the carried partial in registers before the ring starts (vmcnt bookkeeping)
            f32x4 breg[NBUF][CG];
            static_for<0, NBUF - 1>([&](auto c_) {
                constexpr int c = decltype(c_)::value;
                issue_chunk(c, Ab + c * CHF, breg[c]);
            });
            for (int c0 = 0; c0 < nch; c0 += NBUF) {
                static_for<0, NBUF>([&](auto k_) {
                    constexpr int k = decltype(k_)::value;
                    const int c = c0 + k;
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
                    __syncthreads();
                    float* const nbuf = Ab + ((k + NBUF - 1) % NBUF) * CHF;
                    f32x4* const nreg = breg[(k + NBUF - 1) % NBUF];
                    static_assert(NP <= CG * 4, "one piece per k-step");
                    const f32x4* ab = reinterpret_cast<const f32x4*>(Ab + k * CHF) + lane;
                    f32x4 a[2][TPP];
#pragma unroll
                    for (int i = 0; i < TPP; ++i) a[0][i] = ab[i * 64];
                    static_for<0, CG>([&](auto q_) {
                        constexpr int gq = decltype(q_)::value;
                        if constexpr (gq + 1 < CG) {
#pragma unroll
                            for (int i = 0; i < TPP; ++i) a[(gq + 1) & 1][i] = ab[((gq + 1) * PPG + i) * 64];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        static_for<0, 4>([&](auto e_) {
                            constexpr int e = decltype(e_)::value;
                            static_for<0, TPP>([&](auto i_) {
                                constexpr int i = decltype(i_)::value;
                                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[gq & 1][i][e], breg[k][gq][e], acc[i], 0, 0, 0);
                            });
                            if constexpr (gq * 4 + e < NP) {
                                __builtin_amdgcn_sched_barrier(0);
                                issue_piece(std::integral_constant<int, gq * 4 + e>{}, c + NBUF - 1, nbuf, nreg);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        });
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            DL_STAMP(1);
            // ---- BasicLSTMCell pointwise (gate order i, j, f, o; forget_bias 1.0 at run time): the EPI_LSTM expressions
            float* const stg = Ab;                                 // [TPP][16 rows][16 units]
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
                const float hv = dm_tanhf(cc) * so;
                c_reg[i] = cc;
                stg[(i * 16 + rt) * 16 + wave * 4 + uu] = wact ? hv : 0.0f;   // (units beyond H: zeros into the image, they multiply zero weights)
            }
            __syncthreads();
            {
                // the next image: block (tile, group cg), lane L = kq * 16 + r takes row r, units kq + 4e; write-through
                const __amdgpu_buffer_rsrc_t rsN =
                    __builtin_amdgcn_make_buffer_rsrc(him_out + ((size_t)tb * g.hgp + cg) * 256, 0, TPP * g.hgp * 1024, 0x00020000);
#pragma unroll
                for (int q = 0; q < (TPP + 3) / 4; ++q) {
                    const int i = wave + 4 * q;
                    if (i < TPP) {
                        const int r = lane & 15, kq = lane >> 4;
                        u32x4v w4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) w4[e] = __float_as_uint(stg[(i * 16 + r) * 16 + kq + 4 * e]);
                        bstore16_sc1(rsN, w4, (i * g.hgp * 256 + lane * 4) * 4, 0);
                    }
                }
            }
            DL_STAMP(2);
        }
        gs.arrive(tid);
        gs.wait_all((unsigned)(2 * t), wave, lane);
        DL_STAMP(3);
        // =============================================================== phase B: vocabulary pick of step t
        if (bact) {
            f32x4 acc[TMW][TNC];
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int j = 0; j < TNC; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int tvalid = 4 * TPP - wave * TMW;               // tiles of this wave inside the image
            const bool bw = tvalid > 0;                            // (R <= 64: waves 4-7 hold no row tile -- they keep the barriers and leave the matrix pipe to waves 0-3)
            if constexpr (TPP == 1) {
              // ---- R <= 64: loader waves (4-7) + MFMA waves (0-3), kNSTG stages of kKG k-groups
              constexpr int KG = kKG, NSTG = kNSTG, ASTG = 4 * KG * 256, STGF = kB1Stage;
              static_assert(KG % 4 == 0, "W pieces divide over the four loader waves");
              constexpr int WPW = 12 * KG / 16;                        // W pieces (64 float4 each) per loader wave per stage: KG * 16 rows * 12 float4 / 64 / 4
              constexpr int IPW = KG + WPW;
              static_assert((NSTG - 2) * IPW <= 63 && NSTG >= 2, "vmcnt range");
              float* const Sb = smem + kB1Base;
              const int nst = (kg + KG - 1) / KG;
              if (wave >= 4) {
                  const int lw = wave - 4;
                  // image blocks of row tile lw (the image has 4 TPP = 4 tiles), sc1: another XCD wrote them
                  const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(him_out + (size_t)lw * g.hgp * 256, 0, g.hgp * 1024, 0x00020000);
                  auto issue_a = [&](int sidx, int slot) __attribute__((always_inline)) {
                      float* const sb = Sb + slot * STGF;
                      static_for<0, KG>([&](auto gl_) {
                          constexpr int gl = decltype(gl_)::value;
                          const int gi = sidx * KG + gl;
                          const int vo = (gi < kg && sidx < nst) ? lane * 16 : (int)0x80000000u;
                          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsI, (lds_ptr)(sb + (gl * 4 + lw) * 256), 16, vo, (gi < kg ? gi : 0) * 1024, 0, 16);
                      });
                  };
                  // (the W parts of stages 0 .. NSTG-2 were issued during phase A -- they depend on nothing the step computes -- and have long landed)
                  static_for<0, NSTG - 1>([&](auto j_) { issue_a(decltype(j_)::value, decltype(j_)::value); });
                  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * KG) : "memory");
                  __builtin_amdgcn_s_barrier();
                  int slot_n = NSTG - 1;
                  // iteration sidx issues stage sidx + NSTG - 1 (whole) and needs stage sidx + 1 landed: the instructions younger than that are
                  // stages sidx + 2 .. sidx + NSTG - 1, of which those <= NSTG - 2 were image-only (KG instructions) -- peeled so the counts are constants
                  static_for<0, NSTG - 2>([&](auto s_) {
                      constexpr int sidx = decltype(s_)::value;
                      if (sidx < nst) {
                          issue_a(sidx + NSTG - 1, slot_n);
                          issue_w(sidx + NSTG - 1, slot_n);
                          slot_n = slot_n + 1 == NSTG ? 0 : slot_n + 1;
                          constexpr int young = (NSTG - 3 - sidx) * KG + (sidx + 1) * IPW;     // image-only stages sidx+2 .. NSTG-2, whole stages NSTG-1 .. sidx+NSTG-1
                          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(young) : "memory");
                          __builtin_amdgcn_s_barrier();
                      }
                  });
                  DL_STAMP_L(8);                                       // (dev build) loader: everything up to the steady loop
                  for (int sidx = NSTG - 2; sidx < nst; ++sidx) {
                      issue_a(sidx + NSTG - 1, slot_n);
                      issue_w(sidx + NSTG - 1, slot_n);
                      slot_n = slot_n + 1 == NSTG ? 0 : slot_n + 1;
                      DL_STAMP_L(9);                                   // loader: issue
                      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * IPW) : "memory");
                      DL_STAMP_L(10);                                  // loader: data wait
                      __builtin_amdgcn_s_barrier();
                      DL_STAMP_L(11);                                  // loader: barrier (waiting for the MFMA waves)
                  }
              } else {
                  // MFMA waves.  The stage barrier sits BEFORE the last k-group's MFMAs (its fragments are already in registers): behind it the
                  // next stage's first fragments are requested, and the barrier's latency and theirs pass under those twelve MFMAs.
                  __builtin_amdgcn_s_barrier();
                  int slot = 0;
                  f32x4 a4[2];
                  float bvv[2][4][TNC];
                  auto read_g = [&](const float* sb, auto gl_, f32x4& qa, float (&qb)[4][TNC]) __attribute__((always_inline)) {
                      constexpr int gl = decltype(gl_)::value;
                      const f32x4* const ab = reinterpret_cast<const f32x4*>(sb) + wave * 64 + lane;
                      const float* const wb = sb + ASTG + lq * 48 + l15;
                      qa = ab[gl * 4 * 64];
#pragma unroll
                      for (int e = 0; e < 4; ++e)
#pragma unroll
                          for (int j = 0; j < TNC; ++j) qb[e][j] = wb[(gl * 16 + e * 4) * 48 + j * 16];
                  };
                  static_assert(KG % 2 == 0, "the first group of a stage takes register set 0");
                  read_g(Sb, std::integral_constant<int, 0>{}, a4[0], bvv[0]);
                  for (int sidx = 0; sidx < nst; ++sidx) {
                      const float* const sb = Sb + slot * STGF;
                      slot = slot + 1 == NSTG ? 0 : slot + 1;
                      static_for<0, KG>([&](auto gl_) {
                          constexpr int gl = decltype(gl_)::value;
                          if constexpr (gl + 1 < KG) {
                              read_g(sb, std::integral_constant<int, gl + 1>{}, a4[(gl + 1) & 1], bvv[(gl + 1) & 1]);
                          } else {
                              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // every fragment of this stage is in registers
                              __builtin_amdgcn_s_barrier();                              // ... the loaders may refill its slot; the next stage has landed
                              if (sidx + 1 < nst) read_g(Sb + slot * STGF, std::integral_constant<int, 0>{}, a4[0], bvv[0]);
                          }
                          __builtin_amdgcn_sched_barrier(0);
                          static_for<0, 4>([&](auto e_) {
                              constexpr int e = decltype(e_)::value;
#pragma unroll
                              for (int j = 0; j < TNC; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[gl & 1][e], bvv[gl & 1][e][j], acc[0][j], 0, 0, 0);
                          });
                          __builtin_amdgcn_sched_barrier(0);
                      });
                  }
              }
            } else {
              // ---- 257-384 rows (round 6, late): the same split -- waves 4-7 load, waves 0-3 multiply TPP row tiles x 3 column tiles each (the
              // round-3 form had all eight waves load their own image blocks and multiply: 84 us of phase B for 60 us of MFMAs).  One k-group per
              // stage (4 TPP image blocks + 3 KB of weights = 27 KB at 384 rows), kNSTG stages; the ring aliases phase A's LDS (the phases alternate).
              constexpr int NSTG = kNSTG, ASTG = 4 * TPP * 256, STGF = ASTG + 16 * 48;
              static_assert((NSTG - 2) * (TPP + 1) <= 63 && NSTG >= 3, "vmcnt range");
              float* const Sb = smem;
              const int nst = kg;
              if (wave >= 4) {
                  const int lw = wave - 4;
                  const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(him_out + (size_t)lw * TPP * g.hgp * 256, 0, TPP * g.hgp * 1024, 0x00020000);
                  const int q = lw * 64 + lane, brow = q / 12, bc4 = (q % 12) * 4;       // this lane's float4 of the 16 x 48 weight stage (loaders 0-2)
                  auto issue_stage = [&](int sidx, int slot) __attribute__((always_inline)) {
                      float* const sb = Sb + slot * STGF;
                      static_for<0, TPP>([&](auto i_) {
                          constexpr int i = decltype(i_)::value;
                          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsI, (lds_ptr)(sb + (lw * TPP + i) * 256), 16, lane * 16, (i * g.hgp + sidx) * 1024, 0, 16);
                      });
                      if (lw < 3) {
                          const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.Wout + (size_t)sidx * 16 * g.ldwo), 0, 0x7fffffff, 0x00020000);
                          const int k = sidx * 16 + brow;
                          const unsigned off = (k < H && n0 + bc4 < g.V) ? (unsigned)(brow * g.ldwo + n0 + bc4) * 4u : 0x80000000u;
                          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(sb + ASTG + lw * 256), 16, off, 0, 0, 0);
                      }
                  };
                  // (stages beyond the last are NOT issued -- an out-of-range DMA still writes zeros, and the ring is reused for the epilogue's
                  //  accumulator slots: the tail of the loop therefore waits for everything instead of for a count)
#pragma unroll
                  for (int j = 0; j < NSTG - 1; ++j)
                      if (j < nst) issue_stage(j, j);
                  if (lw < 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * (TPP + 1)) : "memory");
                  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * TPP) : "memory");
                  __builtin_amdgcn_s_barrier();
                  int slot_n = NSTG - 1;
                  for (int sidx = 0; sidx < nst; ++sidx) {
                      if (sidx + NSTG - 1 < nst) {
                          issue_stage(sidx + NSTG - 1, slot_n);
                          slot_n = slot_n + 1 == NSTG ? 0 : slot_n + 1;
                          if (lw < 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * (TPP + 1)) : "memory");
                          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * TPP) : "memory");
                      } else {
                          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                      }
                      __builtin_amdgcn_s_barrier();
                  }
              } else {
                  __builtin_amdgcn_s_barrier();
                  f32x4 a4[2][TPP];
                  float bvv[2][4][TNC];
                  auto read_s = [&](int slot, f32x4 (&qa)[TPP], float (&qb)[4][TNC]) __attribute__((always_inline)) {
                      const float* const sb = Sb + slot * STGF;
                      const f32x4* const ab = reinterpret_cast<const f32x4*>(sb) + (wave * TPP) * 64 + lane;
                      const float* const wb = sb + ASTG + lq * 48 + l15;
#pragma unroll
                      for (int i = 0; i < TPP; ++i) qa[i] = ab[i * 64];
#pragma unroll
                      for (int e = 0; e < 4; ++e)
#pragma unroll
                          for (int j = 0; j < TNC; ++j) qb[e][j] = wb[(e * 4) * 48 + j * 16];
                  };
                  read_s(0, a4[0], bvv[0]);
                  int slot = 0;
                  for (int s0 = 0; s0 < nst; s0 += 2) {
                      static_for<0, 2>([&](auto k_) {
                          constexpr int k = decltype(k_)::value;
                          if (s0 + k < nst) {
                              slot = slot + 1 == NSTG ? 0 : slot + 1;
                              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this stage's fragments are in registers
                              __builtin_amdgcn_s_barrier();                              // ... its slot may be refilled; the next stage has landed
                              if (s0 + k + 1 < nst) read_s(slot, a4[(k + 1) & 1], bvv[(k + 1) & 1]);
                              __builtin_amdgcn_sched_barrier(0);
                              static_for<0, 4>([&](auto e_) {
                                  constexpr int e = decltype(e_)::value;
#pragma unroll
                                  for (int i = 0; i < TPP; ++i)
#pragma unroll
                                      for (int j = 0; j < TNC; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[k][i][e], bvv[k][e][j], acc[i][j], 0, 0, 0);
                              });
                              __builtin_amdgcn_sched_barrier(0);
                          }
                      });
                  }
              }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            DL_STAMP(4);
            // ---- the PICK epilogue of gemm_mfma.h (EPI_PICK, two-tier Gumbel-max) on this layout: accumulator (i, j)[r] = row
            // (wave TMW + i) 16 + lq 4 + r, column n0 + 16 j + l15.  The quad l15 = 4q .. 4q+3 holds columns 4Q .. 4Q+3 of the same
            // four rows: quad lane e draws the Philox block of row r = e, words exchanged by DPP.
            const uint32_t e4 = (uint32_t)l15 & 3u;
            unsigned long long* const pk = g.packed + (size_t)t * R * g.pick_stride;
            // The tile loop below stays ROLLED (its body -- three Philox blocks, the two-tier keys, the lane reduction -- is ~1500
            // instructions; unrolled six times it leaves the instruction cache): the accumulators go through a lane-private LDS
            // slot (each lane reads back what it wrote: no barrier), beyond the W stages other waves may still be reading.
            // slot of (tile, column tile j): any wave can finish any tile -- at 257-384 rows the 4 TPP tiles are dealt over all EIGHT waves (the loader
            // waves are idle here), at <= 64 rows waves 0-3 finish their own tile
            f32x4* const zt = reinterpret_cast<f32x4*>(smem + ZA0) + lane;
            if constexpr (TPP != 1) __syncthreads();             // every MFMA wave has read its last fragments, every DMA has landed: the ring is dead
            if (bw)
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int j = 0; j < TNC; ++j) zt[((wave * TMW + i) * TNC + j) * 64] = acc[i][j];
            if constexpr (TPP != 1) __syncthreads();
            constexpr int TSTEP = TPP == 1 ? 4 : 8;              // waves that finish tiles
#pragma nounroll
            for (int tile = wave; tile < (wave < TSTEP ? 4 * TPP : 0); tile += TSTEP) {
                f32x4 ac[TNC];
#pragma unroll
                for (int j = 0; j < TNC; ++j) ac[j] = zt[(tile * TNC + j) * 64];
                const int trow = tile * 16;                             // (uniform) first row of the tile: B is a multiple of 16, so the
                const int tq = __builtin_amdgcn_readfirstlane(trow / g.B);      // tile lies in ONE sample block
                const int mrow = trow + lq * 4;
                // noise ids of the row whose Philox blocks this lane draws (quad lane e4: row lq * 4 + e4), as sampler_rows_kernel
                const int m_own = mrow + (int)e4;
                const int sid_own = m_own < g.noise_rows ? tq : -1;
                const int vid_own = g.video_base + (m_own - tq * g.B);
                bool noisy[4];                                      // sample_id >= 0: the first K B rows
#pragma unroll
                for (int r = 0; r < 4; ++r) noisy[r] = mrow + r < g.noise_rows;
                float best[4];
                uint32_t bidx[4];
                bool have[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { best[r] = 0.0f; bidx[r] = 0xFFFFFFFFu; have[r] = false; }
                float v0[TNC][4], ka[TNC][4];
                uint32_t wd[TNC][4];
#pragma unroll
                for (int j = 0; j < TNC; ++j) {
                    const int col = n0 + j * 16 + l15;
                    u32x4 blk = {0u, 0u, 0u, 0u};
                    if (sid_own >= 0) blk = philox4x32_10((uint32_t)col >> 2, (uint32_t)vid_own, (uint32_t)sid_own, (uint32_t)t, g.seed_lo, g.seed_hi);
                    wd[j][0] = quad_word_from<0>(blk, e4);
                    wd[j][1] = quad_word_from<1>(blk, e4);
                    wd[j][2] = quad_word_from<2>(blk, e4);
                    wd[j][3] = quad_word_from<3>(blk, e4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = mrow + r < R && col < g.V;
                        const float v = ac[j][r] + ep_bias[j];
                        v0[j][r] = v;
                        float k = v;
                        if (noisy[r]) k = v + gumbel_fast_from_word(wd[j][r]);
                        ka[j][r] = ok ? k : -__builtin_inff();
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float mx = ka[0][r];
                    int js = 0;
#pragma unroll
                    for (int j = 1; j < TNC; ++j)
                        if (ka[j][r] > mx) { mx = ka[j][r]; js = j; }
                    float rmx = mx;
#pragma unroll
                    for (int off = 1; off < 16; off <<= 1) rmx = fmaxf(rmx, __shfl_xor(rmx, off, 64));
                    const float thr = rmx - 2.0f * kGumbelScreenMargin;
                    float vs = v0[0][r];
                    uint32_t ws = wd[0][r];
#pragma unroll
                    for (int j = 1; j < TNC; ++j)
                        if (js == j) { vs = v0[j][r]; ws = wd[j][r]; }
                    const int cols = n0 + js * 16 + l15;
                    const bool okr = mrow + r < R;
                    if (okr && cols < g.V && !(mx < thr)) {
                        float v = vs;
                        if (noisy[r]) v = v + gumbel_from_word(ws);
                        v = v + 0.0f;                              // -0 -> +0 so that the integer order equals the float order
                        best[r] = v; bidx[r] = (uint32_t)cols; have[r] = true;
                    }
#pragma unroll
                    for (int j = 0; j < TNC; ++j) {
                        const int col = n0 + j * 16 + l15;
                        const bool extra = okr && col < g.V && j != js && !(ka[j][r] < thr);
                        if (__any(extra)) {
                            if (extra) {
                                float v = v0[j][r];
                                if (noisy[r]) v = v + gumbel_from_word(wd[j][r]);
                                v = v + 0.0f;
                                if (!have[r] || v > best[r] || (v == best[r] && (uint32_t)col < bidx[r])) { best[r] = v; bidx[r] = (uint32_t)col; have[r] = true; }
                            }
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mrow + r;
                    unsigned long long key = have[r] ? (((unsigned long long)orderable(best[r]) << 32) | (uint32_t)(~bidx[r])) : 0ull;
#pragma unroll
                    for (int off = 1; off < 16; off <<= 1) {
                        const unsigned long long o = __shfl_xor(key, off, 64);
                        key = o > key ? o : key;
                    }
                    if (l15 == 0 && m < R && key != 0ull)
                        (void)__hip_atomic_fetch_max(pk + (size_t)m * g.pick_stride, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            DL_STAMP(5);
        }
        if (t + 1 < g.Tc) {
            gs.arrive(tid);
            gs.wait_all((unsigned)(2 * t + 1), wave, lane);
        }
        DL_STAMP(6);
    }
}

typedef void (*DecLoopFn)(const DecLoopArgs);
struct DecLoopCfg { int tpp; DecLoopFn fn; const char* name; };
const DecLoopCfg kDecLoop[] = {{5, decode_loop_kernel<5>, "decloop(m320)"}, {6, decode_loop_kernel<6>, "decloop(m384)"}, {1, decode_loop_kernel<1>, "decloop(m64)"}};
int decloop_lds_bytes(int tpp)
{
    if (tpp == 1) return (kB1Base + kNSTG * kB1Stage) * 4 + 4 * kTNC * 64 * 16;      // phase A region | phase B stages | accumulator slots of waves 0-3 (one row tile x kTNC column tiles each)
    // 257-384 rows: phase A's chunk ring + gate tiles | phase B's stage ring (4 tpp image blocks + 16 x 48 weights per stage), which aliases it and
    // later holds the epilogue's accumulator slots (4 tpp x kTNC KB: smaller than the ring)
    const int a = (kNBUF * kCG * tpp * 256 + 4 * 16 * 20) * 4, b = kNSTG * (4 * tpp * 256 + 16 * 48) * 4;
    return a > b ? a : b;
}
std::once_flag g_dl_once;
bool g_dl_ok = false;
constexpr int kDecLoopGrid = 256;

}  // namespace

// The shape fits, the device has a CU for every workgroup, and the form is switched on for it.  S2VT_DECLOOP: unset / 1 = at <= 64 rows (round 6: the
// loader-wave form of phase B made it the faster one there -- multitask step 4.12 -> 3.94 ms, 57 -> 40 us per decode step); 2 = also at 257-384 rows
// (measured slower than the launches, profiles/NOTES.md); 0 = off.
bool decode_loop_eligible(int R, int H, int E, int V)
{
    static const int on = [] { const char* e = getenv("S2VT_DECLOOP"); return e ? atoi(e) : 1; }();
    const bool rows_ok = on >= 1 && (R <= 64 || (on >= 2 && R > 256 && R <= 384));
    if (!rows_ok || (H & 3) || H < 132 || H > 1008 || E < 1 || (V & 3) || (V + 15) / 16 > kDecLoopGrid * kTNC) return false;
    if (chain_persistent_disabled()) return false;
    ChainHost h;
    if (!chain_host(&h) || h.num_cus < kDecLoopGrid) return false;
    std::call_once(g_dl_once, [] {
        bool ok = true;
        for (const DecLoopCfg& c : kDecLoop) {
            ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(c.fn), hipFuncAttributeMaxDynamicSharedMemorySize, decloop_lds_bytes(c.tpp)) == hipSuccess;
            int nb = 0;
            ok = ok && hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(c.fn), 512, decloop_lds_bytes(c.tpp)) == hipSuccess && nb >= 1;
        }
        g_dl_ok = ok;
    });
    return g_dl_ok;
}

hipError_t launch_decode_loop(const DecLoopLaunch& a, const Dec4Geom& q, hipStream_t st)
{
    ChainHost h;
    if (!chain_host(&h)) return hipErrorInvalidValue;
    DecLoopArgs k;
    std::memset(&k, 0, sizeof(k));
    k.wemb_p = a.wemb_p; k.erow = q.erow; k.w2_p = a.w2_p; k.bias2 = a.bias2;
    k.P2 = a.P2; k.p2_tstride = a.p2_tstride; k.ldp2 = a.ldp2; k.B = a.B; k.c0 = a.c0;
    k.himg0 = a.himg0; k.himg1 = a.himg1; k.packed = a.packed; k.pick_stride = a.pick_stride;
    k.Wout = a.Wout; k.ldwo = a.ldwo; k.bout = a.bout;
    k.seed_lo = (uint32_t)a.seed; k.seed_hi = (uint32_t)(a.seed >> 32); k.noise_rows = a.noise_rows; k.video_base = a.video_base;
    k.R = a.R; k.H = a.H; k.V = a.V; k.Tc = a.Tc;
    k.eg = q.eg; k.hg = q.hg; k.hgp = q.hgp; k.ncg = q.ncg; k.ech = q.ech; k.hch = q.hch;
    k.sync = a.sync; k.status = h.status_dev; k.fault = h.fault; k.spin_limit = h.spin_limit;
    const DecLoopCfg& c = kDecLoop[q.tpp == 5 ? 0 : (q.tpp == 6 ? 1 : 2)];
    ChainLaunchOrder order;                                     // one persistent grid at a time per process
    hipError_t e = order.before(st, h.device);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(a.sync, 0, kChainSyncBytes, st);
    if (e != hipSuccess) return e;
    const double flops = (2.0 * a.R * (double)(a.E + a.H) * 4.0 * a.H + 2.0 * a.R * (double)a.H * a.V) * a.Tc;
    const int ci = 14 + (q.tpp == 5 ? 0 : (q.tpp == 6 ? 1 : 2));                   // profiler slot: class 2 (vocabulary pick), beyond the gemm_kernel table
    if (!prof_wants(2, ci)) {
        hipLaunchKernelGGL(c.fn, dim3(kDecLoopGrid), dim3(512), decloop_lds_bytes(c.tpp), st, k);
        e = hipGetLastError();
        return e != hipSuccess ? e : order.after(st, h.device);
    }
    hipEvent_t e0, e1;
    hipError_t pe = prof_events(&e0, &e1);
    if (pe != hipSuccess) return pe;
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL(c.fn, dim3(kDecLoopGrid), dim3(512), decloop_lds_bytes(c.tpp), st, k);
    (void)hipEventRecord(e1, st);
    prof_record(2, ci, c.name, flops, e0, e1);
    e = hipGetLastError();
    return e != hipSuccess ? e : order.after(st, h.device);
}

}  // namespace s2vt

#ifdef S2VT_DL_STAMP
extern "C" int s2vt_dl_stamp_read(unsigned long long* out16)
{
    if (hipDeviceSynchronize() != hipSuccess) return -4;
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(s2vt::dl_stamp_acc), 16 * sizeof(unsigned long long)) != hipSuccess) return -4;
    unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(s2vt::dl_stamp_acc), z, sizeof(z)) == hipSuccess ? 0 : -4;
}
#endif
