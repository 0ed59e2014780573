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
    static_assert(TN % NG == 0, "TN must split evenly over the column groups");
};

template <int WM, int WN, int TM, int TN, int NG, int EPI, bool VEC>
__global__ __launch_bounds__(64 * WM * WN) void gemm_kernel(const GemmArgs g)
{
    using Cfg = GemmCfg<WM, WN, TM, TN, NG, EPI, VEC>;
    constexpr int NT = Cfg::NT, BM = Cfg::BM, BN = Cfg::BN, TNG = Cfg::TNG, CG = Cfg::CG, SB = Cfg::SB;
    constexpr int A4 = Cfg::A4, B4 = Cfg::B4;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][BM][SA]
    float* Bs = smem + 2 * BM * SA;         // [2][BK][SB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, lq = lane >> 4;

    // blockIdx.x walks the column tiles fastest, so workgroups that share an A row-block are launched
    // together, and the 8 XCD L2s each see every 8th column tile of the weight matrix.
    const int ntile_n = (g.N + CG - 1) / CG;
    const int tile_n = blockIdx.x % ntile_n;
    const int tile_m = blockIdx.x / ntile_n;
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

    // per-thread staging slots
    float4 ra[A4], rb[B4];
    const float* arow[A4];

    const int kbeg = g.splits > 1 ? (int)blockIdx.y * g.kper : 0;
    auto seg_len = [&](int sidx) {
        const int k = g.seg[sidx].k - kbeg;
        return g.splits > 1 ? (k < g.kper ? k : g.kper) : k;
    };

    int s = 0;
    while (s < g.nseg && (g.seg[s].ptr == nullptr || seg_len(s) <= 0)) ++s;
    int kc = 0;  // chunk offset inside segment s

    auto seg_rows = [&](int sidx) {
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            const int idx = tid + i * NT;
            const int r = idx / (BK / 4);
            int m = m0 + r;
            const ASeg& sg = g.seg[sidx];
            const float* p = nullptr;
            if (idx < BM * (BK / 4) && m < g.M) {
                if (sg.rowmod > 0) m %= sg.rowmod;
                if (sg.rowidx) m = sg.rowidx[m];
                if (sg.rowkey) m = (int)(~(uint32_t)sg.rowkey[m]);
                p = sg.ptr + (size_t)m * sg.ld + kbeg;
            }
            arow[i] = p;
        }
    };

    auto load_chunk = [&](int sidx, int koff) {
        const ASeg& sg = g.seg[sidx];
        const int sk = seg_len(sidx);
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            const int idx = tid + i * NT;
            const int k = koff + (idx % (BK / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (arow[i]) {
                if (VEC) {
                    if (k < sk) v = *reinterpret_cast<const float4*>(arow[i] + k);
                } else {
                    if (k + 0 < sk) v.x = arow[i][k + 0];
                    if (k + 1 < sk) v.y = arow[i][k + 1];
                    if (k + 2 < sk) v.z = arow[i][k + 2];
                    if (k + 3 < sk) v.w = arow[i][k + 3];
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B4; ++i) {
            const int idx = tid + i * NT;
            const int kr = idx / (BN / 4);
            const int c = (idx % (BN / 4)) * 4;
            const int grp = c / CG, cc = n0 + c % CG;
            const int k = koff + kr;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < BK * (BN / 4) && k < sk) {
                const float* wp = g.W + (size_t)(sg.kw + kbeg + k) * g.ldw + grp * g.gstride + cc;
                if (VEC) {
                    if (cc < g.N) v = *reinterpret_cast<const float4*>(wp);
                } else {
                    if (cc + 0 < g.N) v.x = wp[0];
                    if (cc + 1 < g.N) v.y = wp[1];
                    if (cc + 2 < g.N) v.z = wp[2];
                    if (cc + 3 < g.N) v.w = wp[3];
                }
            }
            rb[i] = v;
        }
    };

    auto store_chunk = [&](int buf) {
        float* a = As + buf * BM * SA;
        float* b = Bs + buf * BK * SB;
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            const int idx = tid + i * NT;
            if (idx < BM * (BK / 4)) {
                float* d = a + (idx / (BK / 4)) * SA + (idx % (BK / 4)) * 4;
                *reinterpret_cast<float2*>(d) = make_float2(ra[i].x, ra[i].y);
                *reinterpret_cast<float2*>(d + 2) = make_float2(ra[i].z, ra[i].w);
            }
        }
#pragma unroll
        for (int i = 0; i < B4; ++i) {
            const int idx = tid + i * NT;
            if (idx < BK * (BN / 4)) {
                float* d = b + (idx / (BN / 4)) * SB + (idx % (BN / 4)) * 4;
                *reinterpret_cast<float4*>(d) = rb[i];
            }
        }
    };

    auto advance = [&](int& sidx, int& koff) {  // next chunk position; returns via refs
        koff += BK;
        if (koff >= seg_len(sidx)) {
            koff = 0;
            ++sidx;
            while (sidx < g.nseg && (g.seg[sidx].ptr == nullptr || seg_len(sidx) <= 0)) ++sidx;
        }
    };

    if (s < g.nseg) {
        seg_rows(s);
        load_chunk(s, 0);
        store_chunk(0);
    }
    __syncthreads();

    int buf = 0;
    while (s < g.nseg) {
        int ns = s, nk = kc;
        advance(ns, nk);
        const bool more = ns < g.nseg;
        if (more) {
            if (ns != s) seg_rows(ns);
            load_chunk(ns, nk);
        }
        const float* a = As + buf * BM * SA + ((wm * TM) * 16 + l15) * SA + lq;
        const float* b = Bs + buf * BK * SB + lq * SB + l15;
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
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
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
        buf ^= 1;
        s = ns;
        kc = nk;
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
