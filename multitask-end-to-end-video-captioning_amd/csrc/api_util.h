// api_util.h -- host-side helpers shared by the extern "C" translation units (api.hip, train.hip)
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/s2vt.h"
#include "internal.h"

using namespace s2vt;

namespace s2vt_api {

extern std::atomic<int> g_last_hip;

inline int hip_fail(hipError_t e)
{
    g_last_hip.store((int)e);
    return S2VT_E_HIP;
}

#define HIP_TRY(expr)                               \
    do {                                            \
        hipError_t _e = (expr);                     \
        if (_e != hipSuccess) return hip_fail(_e);  \
    } while (0)

inline hipStream_t S(s2vt_stream s) { return reinterpret_cast<hipStream_t>(s); }

// bump allocator over the caller's workspace (256-byte granules)
struct Carver {
    char* base;
    size_t off = 0, cap;
    Carver(void* p, size_t bytes) : base(static_cast<char*>(p)), cap(bytes) {}
    template <typename T>
    T* take(size_t n)
    {
        const size_t bytes = (n * sizeof(T) + 255) & ~size_t(255);
        T* r = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += bytes;
        return r;
    }
    bool ok() const { return off <= cap; }
};

inline void seg_from_operand(ASeg& s, const s2vt_operand* o, int kw)
{
    std::memset(&s, 0, sizeof(s));
    if (!o) return;
    s.ptr = o->ptr;
    s.rowidx = o->rowidx;
    s.ld = o->ld;
    s.k = o->k;
    s.kw = kw;
    s.rowmod = o->rowmod;
}

// The optional second stream of the backward passes (defined in train.hip; one per DEVICE, picked by hipGetDevice at the call): weight-gradient contractions that do not feed a
// recurrence run there BESIDE the persistent backward recurrence they are independent of, released by a gate once the recurrence's grid is
// resident (internal.h ChainGate).  S2VT_OVERLAP: 2 (default) gated, 1 the ungated round-1 form (S2VT path only), 0 one stream.
struct SideStream {
    hipStream_t s = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ok = false;
    int mode = 0;
};
SideStream& side_stream();
std::mutex& side_stream_mutex();     // held while a call ENQUEUES work that uses the side stream and its events (one set per device; the one mutex serialises enqueueing, not execution)
hipError_t fork_to(hipStream_t from, hipStream_t to, hipEvent_t ev);      // `to` waits for everything issued so far on `from`

constexpr int kPickStride = 16;      // sampler workspace: one packed pick per 128-byte line (GemmArgs::pick_stride)

inline ASeg make_seg(const float* ptr, int ld, int k, int kw, int rowmod = 0, const int* rowidx = nullptr,
              const unsigned long long* rowkey = nullptr, int rowkey_stride = 1)
{
    ASeg s;
    std::memset(&s, 0, sizeof(s));
    s.ptr = ptr; s.ld = ld; s.k = k; s.kw = kw; s.rowmod = rowmod; s.rowidx = rowidx; s.rowkey = rowkey; s.rowkey_stride = rowkey_stride;
    return s;
}

struct NoiseIds {
    const int32_t* video_id;
    const int32_t* sample_id;
    uint64_t seed;
};

// One BasicLSTMCell call.  segs/nseg describe [x0 ; x1 ; h_prev] with their W row offsets.
inline hipError_t lstm_call(const ASeg* segs, int nseg, const float* W, const float* b, const float* c_prev, int cprev_rowmod,
                     float* c_new, float* h_new, float* out, float* gates, int M, int H, float keep, const NoiseIds& ids,
                     uint32_t drop_code, int cfg, hipStream_t st, const float* cinit = nullptr, int ldcinit = 0,
                     int cinit_rowmod = 0, const int* omap = nullptr, const int* m_dev = nullptr)
{
    GemmArgs a;
    std::memset(&a, 0, sizeof(a));
    for (int i = 0; i < nseg; ++i) a.seg[i] = segs[i];
    a.nseg = nseg;
    a.W = W; a.ldw = 4 * H; a.M = M; a.N = H; a.gstride = H; a.bias = b;
    a.c_prev = c_prev; a.cprev_rowmod = cprev_rowmod; a.c_new = c_new; a.h_new = h_new; a.out = out; a.gates = gates;
    a.keep = keep; a.drop_code = drop_code;
    a.video_id = ids.video_id; a.sample_id = ids.sample_id;
    a.seed_lo = (uint32_t)ids.seed; a.seed_hi = (uint32_t)(ids.seed >> 32);
    a.cinit = cinit; a.ldcinit = ldcinit; a.cinit_rowmod = cinit_rowmod;   // a carried partial chain (hoisted input products)
    a.omap = omap; a.m_dev = m_dev;                                          // live-row launch (gemm_mfma.h)
    return launch_gemm(a, EPI_LSTM, cfg, st);
}

inline hipError_t pick_call(const float* A, int lda, const float* W, const float* b, int M, int H, int V, const NoiseIds& ids,
                     int step, unsigned long long* packed, float* logits_out, int cfg, hipStream_t st, int pick_stride = 1,
                     const int* omap = nullptr, const int* m_dev = nullptr)
{
    GemmArgs a;
    std::memset(&a, 0, sizeof(a));
    a.seg[0] = make_seg(A, lda, H, 0);
    a.nseg = 1;
    a.W = W; a.ldw = V; a.M = M; a.N = V; a.gstride = 0; a.bias = b;
    a.video_id = ids.video_id; a.sample_id = ids.sample_id;
    a.seed_lo = (uint32_t)ids.seed; a.seed_hi = (uint32_t)(ids.seed >> 32);
    a.step = step; a.pick = packed; a.pick_stride = pick_stride; a.logits_out = logits_out; a.ldc = V;
    a.omap = omap; a.m_dev = m_dev;
    return launch_gemm(a, EPI_PICK, cfg, st);
}

inline hipError_t store_call(const ASeg* segs, int nseg, const float* W, int ldw, const float* bias, float* C, int ldc, int M,
                      int N, int act, int cfg, hipStream_t st, const float* cinit = nullptr, int ldcinit = 0, bool w_transposed = false)
{
    GemmArgs a;
    std::memset(&a, 0, sizeof(a));
    for (int i = 0; i < nseg; ++i) a.seg[i] = segs[i];
    a.nseg = nseg;
    a.W = W; a.ldw = ldw; a.M = M; a.N = N; a.gstride = 0; a.bias = bias;
    a.cinit = cinit; a.ldcinit = ldcinit;
    a.C = C; a.ldc = ldc; a.act = act;
    return launch_gemm(a, w_transposed ? EPI_STORE_NT : (int)EPI_STORE, cfg, st);
}

// ---- order-free data-gradient products (W^T form) with optional split-K slabs: shared by train.hip and attn_model.hip
// The same for the other two data-gradient products of the backward when they are short of tiles: dX2 = dZ2 @ W2[0:H+E]^T
// ([T N, H+E], K = 4H) and dX1 = dZ1 @ W1[0:E]^T ([Tv B, E], K = 4H).  One slab buffer serves both (they run one after the other).
constexpr int kDxSlabRows = 2048, kDxMaxSlabs = 12;
// Tile and slab count of such a product, from a sweep over 384..1600 rows x {dO2, dX2} x six tiles x seven slab counts
// (tools/tune_slabs.py): the 64x64 tile (four waves along the rows) wins or ties everywhere -- slabs of ~1500 reduction steps
// spend a third of their time in prologue and epilogue, which many small co-resident workgroups overlap and two big ones
// per CU cannot -- with slabs of ~1536 steps and at least ~768 workgroups: dX2 at 1216 rows 213 -> 142 us, at 640 rows
// 130 -> 83; dO2 at 384 rows 169 -> 95, at 896 rows 216 -> 185.
constexpr int kSlabTileCfg = 0;        // kStoreNT[0] = nt64x64(4x1)
inline int slab_splits(int M, int N, int K, int max_slabs)
{
    const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
    long s = (K + 1535) / 1536;
    const long fill = (768 + tiles - 1) / tiles;
    if (fill > s) s = fill;
    if (s > max_slabs) s = max_slabs;
    while (s > 1 && K / s < 256) --s;                                            // keep >= 8 chunks per slab
    return s < 1 ? 1 : (int)s;
}
inline int dx_splits(int M, int N, int K)
{
    static const bool off = [] { const char* e = getenv("S2VT_DX_SPLITS"); return e && e[0] == '0'; }();      // dev knob
    if (off || M > kDxSlabRows || ((size_t)M * N & 3)) return 1;
    return slab_splits(M, N, K, kDxMaxSlabs);
}
// order-free product for the backward data path: C[s] = A[:, Ks] @ Wt[:, Ks]^T with Wt = the FORWARD weight block as it
// lies in memory ([N rows][K columns], row stride ldw) -- no transposed copies (split-K slabs when splits > 1)
inline hipError_t nn_bwd(const float* A, int lda, const float* Wt, int ldw, float* C, int ldc, int M, int N, int K, int splits,
                  size_t slab_stride, hipStream_t st, int tile_cfg = -1)
{
    GemmArgs a;
    std::memset(&a, 0, sizeof(a));
    a.seg[0] = make_seg(A, lda, K, 0);
    a.nseg = 1;
    a.W = Wt; a.ldw = ldw; a.M = M; a.N = N; a.C = C; a.ldc = ldc;
    if (splits > 1) {
        a.splits = splits;
        a.kper = ((K + splits - 1) / splits + BK - 1) / BK * BK;
        a.splits = (K + a.kper - 1) / a.kper;
        a.slab_stride = slab_stride;
    }
    static const int cfg_ = [] { const char* e = getenv("S2VT_SLAB_CFG"); return e ? atoi(e) : -1; }();     // dev knob
    return launch_gemm(a, EPI_STORE_NT, tile_cfg >= 0 ? tile_cfg : (splits > 1 ? cfg_ : -1), st);
}

// C = A @ Wt^T with the reduction cut into slabs when the output is short of tiles (dx_splits), the slabs summed into C
inline hipError_t nn_bwd_slabs(const float* A, int lda, const float* Wt, int ldw, float* C, int ldc, int M, int N, int K, float* slabs, hipStream_t st,
                        size_t slab_floats = ~(size_t)0)
{
    int s = slabs ? dx_splits(M, N, K) : 1;
    if ((size_t)s * M * N > slab_floats) s = 1;                                  // (a row count the carve did not see)
    if (s <= 1 || ldc != N) return nn_bwd(A, lda, Wt, ldw, C, ldc, M, N, K, 1, 0, st);
    const size_t stride = (size_t)M * N;
    hipError_t e = nn_bwd(A, lda, Wt, ldw, slabs, N, M, N, K, s, stride, st, kSlabTileCfg);
    if (e != hipSuccess) return e;
    const int kper = ((K + s - 1) / s + BK - 1) / BK * BK;                     // what nn_bwd made of `splits`
    return launch_sum_slabs(C, slabs, (K + kper - 1) / kper, stride, stride, st);
}


// ---- sampler halves (api.hip), shared with the session API (session.hip)
struct SampleWs {
    float *emb, *Xp1, *c1, *h1, *G1, *P2, *c2e, *h2e, *c2[2], *h2[2];   // c2e / h2e: LSTM2 state history of the encoding stage [Tv+1][B][H]
    unsigned long long* packed;
    int32_t *vid, *sid, *bos;
    float* chain_abuf;       // persistent-recurrence scratch (chain.hip): fragment images of h + arrival counters
    unsigned* chain_sync;
    float *wemb_p, *w2_p, *himg[2];   // fragment-order operands of the decode loop's LSTM2 step (decode4.hip); NULL when R is outside its range
    int32_t* live[2];        // stop-at-<eos> mode: the rows still sampling at the current / next step (ascending), ...
    int32_t* nlive;          // ... and their count per step [Tc + 1], device-resident
};

// One LSTM recurrence of T steps on M rows: ONE persistent launch when the shape fits (chain_eligible), else T
// per-step launches of the fused cell kernel.  Histories: step t reads slot t of C / Hh and writes slot t + 1.
hipError_t lstm_recurrence(const float* W, int kw0, const float* bias, const float* cinit, size_t cinit_tstride, int ldcinit,
                           int cinit_steps, float* C, float* Hh, size_t state_tstride, float* gates, size_t gates_tstride,
                           float* out, size_t out_tstride, int M, int H, int T, float keep, const NoiseIds& ids,
                           uint32_t drop_code0, float* chain_abuf, unsigned* chain_sync, hipStream_t st, const int32_t* perm = nullptr,
                           const int32_t* nlive = nullptr);
size_t carve_sample(Carver& c, const s2vt_dims* d, int B, int R, SampleWs* w);
int sample_encode(const s2vt_dims* d, const s2vt_params* p, const float* video, int B, const SampleWs& w, s2vt_stream stream);
int sample_decode(const s2vt_dims* d, const s2vt_params* p, int B, int K, int with_greedy, uint64_t seed, int video_base,
                  int32_t* ids_out, const SampleWs& w, s2vt_stream stream, int stop_at_eos = 0);
bool sampler_params_ok(const s2vt_params* p);

inline bool dims_ok(const s2vt_dims* d)
{
    return d && d->dim_image > 0 && d->n_words > 0 && d->word_dim > 0 && d->lstm_dim > 0 && d->n_video_lstm_step > 0 &&
           d->n_caption_lstm_step > 0;
}


}  // namespace s2vt_api
