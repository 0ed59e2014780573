// train.hip -- extern "C" entry points of the update half of the step: teacher-forced unroll with
// saved activations (build_model / build_loss), softmax-NLL forward+backward (XE and the
// reward-scaled policy-gradient NLL), back-propagation through time, gradient finalisation,
// global-norm clip + TF-form Adam.
#include <hip/hip_runtime.h>

#include <functional>
#include <mutex>

#include <cstdlib>

#include "api_util.h"

using namespace s2vt_api;

namespace s2vt_api {
// An optional second stream for the backward pass: the recurrences (25 dependent steps of small kernels, ~half the
// matrix pipes idle) run on the caller's stream while weight-gradient contractions that do not depend on them
// run here, forked / joined with events so the call keeps its stream semantics.  Created once per process.
// (mode 1: S2VT_OVERLAP=1, ungated, below; 2: gated overlap with the persistent backward recurrences, round 5)
// One per DEVICE (the first version was a process-wide static created on whichever device was current at first use -- a second model
// on another GPU then recorded / waited on device 0's events): indexed by hipGetDevice like chain.hip's per-device state, each entry created on
// first use with its device current.  A device index beyond the table, or a creation failure, yields the one-stream entry (ok = false).
SideStream& side_stream()
{
    constexpr int kMaxDev = 32;
    static SideStream table[kMaxDev];
    static bool made[kMaxDev] = {};
    static SideStream none;                              // ok = false, mode 0
    static const int mode = [] {
        // Mode 1 (S2VT_OVERLAP=1, ungated).  Measured on MI355X: the two streams do run concurrently, but the kernels
        // only slow each other down (TN 724 -> 1462 us, slab GEMM 28 -> 46 us per launch) for a net 0.1 ms of
        // 15.6, and per-launch durations stop meaning anything for the roofline.
        // Mode 2 (round 5, the default; S2VT_OVERLAP=0 switches it off): the weight-gradient contractions that do not feed a recurrence run
        // on the side stream BESIDE the persistent backward recurrence they are independent of -- dWout beside LSTM2's, LSTM2's three beside
        // LSTM1's -- released by a gate once the recurrence's grid is resident (internal.h ChainGate).  A persistent recurrence at <= 256 rows
        // is one wave per SIMD at <= 380 VGPRs that waits in hand-offs half the time; the contraction fills the other half of the pipe.
        const char* on = getenv("S2VT_OVERLAP");
        const int m = on ? atoi(on) : 2;
        return (m < 1 || m > 2) ? 0 : m;
    }();
    int dev = -1;
    if (mode == 0 || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return none;
    std::lock_guard<std::mutex> lk(side_stream_mutex());
    if (!made[dev]) {
        made[dev] = true;
        SideStream& t = table[dev];
        t.mode = mode;
        bool good = hipStreamCreateWithFlags(&t.s, hipStreamNonBlocking) == hipSuccess;
        for (auto& e : t.ev)
            good = good && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        t.ok = good;
    }
    return table[dev];
}
// side waits for everything issued so far on `from`
hipError_t fork_to(hipStream_t from, hipStream_t to, hipEvent_t ev)
{
    hipError_t e = hipEventRecord(ev, from);
    return e != hipSuccess ? e : hipStreamWaitEvent(to, ev, 0);
}
std::mutex& side_stream_mutex()
{
    static std::mutex mu;
    return mu;
}
}  // namespace s2vt_api

namespace {

__global__ void enc_index_kernel(int32_t* idx, int B, int Tv)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Tv) return;
    const int t = i / B, j = i % B;
    idx[i] = j * Tv + t;     // row of video[B*Tv, d] / emb[B*Tv, E] feeding (step t, video j)
}

// Saved activations + backward scratch of one teacher-forced unroll, carved from the caller's buffer.
// LSTM1 never sees the caption (its input is the frame embedding, then the zero padding), so its clean
// state trajectory is per VIDEO: it is computed and back-propagated on B rows, and only its
// dropout-wrapped output O1 is expanded to the N = rep*B sample rows.
struct TrainWs {
    float *emb, *Xp1;
    int32_t *prev, *tgt, *encidx;
    float *G1, *C1, *H1, *O1, *G2, *C2, *H2, *O2;
    float *dO2, *dZ1, *dZ2, *dX2, *dX1, *dH1, *slab, *dc;
    int32_t* decidx;     // inverse of encidx: row of dX1 (time-major) for row j*Tv + t of d_video
    float* chain_abuf;   // persistent-recurrence scratch (chain.hip)
    unsigned* chain_sync;
    float *bimg, *bex;   // persistent BACKWARD recurrence scratch (chain.hip): dz images, partial-tile exchange
    unsigned* bsync;
    float* dO2s;         // split-K slabs of dO2 = dlogits @ Wout^T when it has few rows (NULL otherwise)
    size_t dO2s_rows;    // ... rows x slabs it holds
    float* dO2p;         // dO2 of the LIVE rows only (the *_live entry points), scattered into dO2
    float *dZ2p, *dX2p;  // the live decode rows of dZ2 (packed copy) and of dX2 (computed packed, scattered into dX2)
    int32_t* prevp;      // previous word of the live decode rows
    int32_t *perm, *nlive;   // live rows in the recurrences (chain_live_capable shapes): LSTM2's rows by length, live rows per step
    size_t dXs_floats;   // capacity of dXs
    float* dXs;          // split-K slabs of dX2 / dX1 when they are short of tiles (NULL otherwise)
};

// Split-K plan of the recurrent data-gradient product dz[M,4H] @ Whh^T[4H,H] (order-free): enough K slabs
// that the launch has >= ~512 workgroups of the 64x32 tile; the slabs are summed by the next step's
// pointwise kernel.
constexpr int kMaxSlabs = 16;
// dO2 = dlogits[Tc N, V] @ Wout^T -> [Tc N, H] has K = |V| = 12000 but, at N = 64, only 1280 x 1000 outputs: as 64x32 tiles with
// the whole K each it ran 80 TFLOP/s (12 % of the XE step).  Order-free, so up to this many rows it is cut into K slabs
// and the slabs are summed.
constexpr int kDo2SplitRows = 2048, kDo2Slabs = 12;
int do2_splits(int rows, int H, int V)
{
    static const int cap = [] { const char* e = getenv("S2VT_DO2_SPLITS"); const int v = e ? atoi(e) : kDo2Slabs; return v < 1 ? 1 : (v > kDo2Slabs ? kDo2Slabs : v); }();   // dev knob
    if (rows > kDo2SplitRows || (H & 3)) return 1;
    return slab_splits(rows, H, V, cap);
}
struct SlabPlan { int splits, kper, nslab; };
SlabPlan slab_plan(int M, int H)
{
    const int K = 4 * H;
    static const int want = [] { const char* e = getenv("S2VT_SLAB_WGS"); return e ? atoi(e) : 512; }();     // dev knob
    static const int tn_ = [] { const char* e = getenv("S2VT_SLAB_TILE_N"); return e ? atoi(e) : 32; }();   // dev knob (columns of the tile nn_bwd will get)
    const long tiles = (long)((M + 63) / 64) * ((H + tn_ - 1) / tn_);
    int splits = (int)((want + tiles - 1) / tiles);
    if (splits < 1) splits = 1;
    if (splits > kMaxSlabs) splits = kMaxSlabs;
    SlabPlan p;
    p.kper = ((K + splits - 1) / splits + BK - 1) / BK * BK;      // what nn_bwd computes from `splits`
    p.nslab = (K + p.kper - 1) / p.kper;
    p.splits = splits;
    return p;
}

// scratch of the persistent backward recurrence for any row count up to Mmax (the one- and two-part forms lay it out differently)
void bwd_scratch_max(int H, int Mmax, size_t* imgf, size_t* exf, size_t* syncb)
{
    *imgf = *exf = *syncb = 0;
    for (int m : {64, 128, 256, 384}) {
        size_t a, b, s;
        bwd_chain_scratch(H, Mmax < m ? Mmax : m, &a, &b, &s);
        if (a > *imgf) *imgf = a;
        if (b > *exf) *exf = b;
        if (s > *syncb) *syncb = s;
    }
}

size_t carve_train(Carver& c, const s2vt_dims* d, int B, int N, TrainWs* out)
{
    const size_t H = d->lstm_dim, E = d->word_dim, Tv = d->n_video_lstm_step, Tc = d->n_caption_lstm_step;
    const size_t T = Tv + Tc, n = N, b = B;
    TrainWs w;
    w.emb = c.take<float>(b * Tv * E);
    w.Xp1 = c.take<float>(b * Tv * 4 * H);
    w.prev = c.take<int32_t>(Tc * n); w.tgt = c.take<int32_t>(Tc * n); w.encidx = c.take<int32_t>(Tv * b);
    w.G1 = c.take<float>(T * b * 4 * H); w.C1 = c.take<float>((T + 1) * b * H); w.H1 = c.take<float>((T + 1) * b * H);
    w.O1 = c.take<float>(T * n * H);
    w.G2 = c.take<float>(T * n * 4 * H); w.C2 = c.take<float>((T + 1) * n * H); w.H2 = c.take<float>((T + 1) * n * H);
    w.O2 = c.take<float>(T * n * H);
    w.dO2 = c.take<float>(Tc * n * H);
    w.dO2p = c.take<float>(Tc * n * H);
    w.dZ2p = c.take<float>(Tc * n * 4 * H); w.dX2p = c.take<float>(Tc * n * (H + E)); w.prevp = c.take<int32_t>(Tc * n);
    w.dZ1 = c.take<float>(T * b * 4 * H); w.dZ2 = c.take<float>(T * n * 4 * H);
    w.dX2 = c.take<float>(T * n * (H + E)); w.dX1 = c.take<float>(Tv * b * E); w.dH1 = c.take<float>(T * b * H);
    w.slab = c.take<float>((size_t)kMaxSlabs * n * H); w.dc = c.take<float>(n * H);
    w.decidx = c.take<int32_t>(Tv * b);
    w.perm = c.take<int32_t>(n); w.nlive = c.take<int32_t>(T);
    w.chain_sync = c.take<unsigned>(kChainSyncBytes / 4);
    w.chain_abuf = c.take<float>(chain_scratch_floats((int)H));
    {
        size_t imgf, exf, syncb;
        bwd_scratch_max((int)H, N, &imgf, &exf, &syncb);       // (sized for the larger of the two recurrences: N >= B rows)
        w.bimg = c.take<float>(imgf); w.bex = c.take<float>(exf); w.bsync = c.take<unsigned>(syncb / 4);
    }
    // (a truncated unroll -- caption_steps < Tc, the *_steps entry points -- has fewer rows and may split where the full one
    //  does not: both scratch blocks are sized for the worst of all step counts)
    {
        size_t need = 0;
        for (size_t tc = 1; tc <= Tc; ++tc) {
            const int s2 = do2_splits((int)(tc * n), (int)H, d->n_words);
            if (s2 > 1 && (size_t)s2 * tc * n > need) need = (size_t)s2 * tc * n;
        }
        w.dO2s = need ? c.take<float>(need * H) : nullptr;
        w.dO2s_rows = need;
    }
    {
        size_t need = 0;
        for (size_t tc = 1; tc <= Tc; ++tc) {
            const int s2 = dx_splits((int)((Tv + tc) * n), (int)(H + E), (int)(4 * H));
            const size_t need2 = s2 > 1 ? (size_t)s2 * (Tv + tc) * n * (H + E) : 0;
            if (need2 > need) need = need2;
        }
        const int s1 = dx_splits((int)(Tv * b), (int)E, (int)(4 * H));
        const size_t need1 = s1 > 1 ? (size_t)s1 * Tv * b * E : 0;
        if (need1 > need) need = need1;
        w.dXs = need ? c.take<float>(need) : nullptr;
        w.dXs_floats = need;
    }
    if (out) *out = w;
    return c.off;
}

bool params_ok(const s2vt_params* p)
{
    return p && p->Wemb && p->encode_image_W && p->encode_image_b && p->lstm1_W && p->lstm1_b && p->lstm2_W && p->lstm2_b &&
           p->embed_word_W && p->embed_word_b;
}

// Back-propagation through one cell's unroll: dZ[t] for t = T-1 .. 0.  ONE persistent launch when the shape fits
// (bwd_chain_eligible: M <= 128 rows by default), else per step {pointwise, split-K slabs of dz @ Whh^T summed by the next
// pointwise launch}.  dext: upstream gradient w.r.t. the (dropped) output of step t >= dext_t0.
struct BwdScratch { float* slab; float* dc; float* bimg; float* bex; unsigned* bsync; };
hipError_t lstm_recurrence_bwd(const float* W, int kw0, const float* gates, const float* C, const float* dext, size_t dext_tstride,
                               int ld_ext, int dext_t0, float* dZ, int M, int H, int T, float keep, uint64_t seed, uint32_t drop_code0,
                               const int32_t* video_id, const int32_t* sample_id, const BwdScratch& sc, int persistent, hipStream_t st,
                               const int32_t* perm = nullptr, const int32_t* nlive = nullptr)
{
    const size_t MH = (size_t)M * H;
    const bool can = sc.bimg && sc.bex && sc.bsync && bwd_chain_eligible(M, H) && !(reinterpret_cast<uintptr_t>(W) & 15);
    if (persistent == 1 && !can) return hipErrorInvalidValue;
    if (persistent == 1 || (persistent == -1 && can && bwd_chain_auto(M, H))) {
        BwdChainLaunch a;
        std::memset(&a, 0, sizeof(a));
        a.W = W; a.ldw = 4 * H; a.kw0 = kw0; a.gates = gates; a.gates_tstride = 4 * MH; a.C = C; a.state_tstride = MH;
        a.dext = dext; a.dext_tstride = dext_tstride; a.ld_ext = ld_ext; a.dext_t0 = dext_t0;
        a.dZ = dZ; a.dz_tstride = 4 * MH; a.M = M; a.H = H; a.T = T;
        a.keep = keep; a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.drop_code0 = drop_code0;
        a.video_id = video_id; a.sample_id = sample_id;
        a.img = sc.bimg; a.ex = sc.bex; a.sync = sc.bsync;
        a.perm = perm; a.nlive = nlive;
        return launch_lstm_bwd_chain(a, st);
    }
    const SlabPlan sp = slab_plan(M, H);
    for (int t = T - 1; t >= 0; --t) {
        hipError_t e = launch_lstm_bwd_pointwise(gates + (size_t)t * 4 * MH, C + (t + 1) * MH, C + t * MH, t == T - 1 ? nullptr : sc.slab, sp.nslab, MH,
                                                 (dext && t >= dext_t0) ? dext + (size_t)(t - dext_t0) * dext_tstride : nullptr, ld_ext,
                                                 t == T - 1 ? nullptr : sc.dc, sc.dc, dZ + (size_t)t * 4 * MH, M, H, keep, seed, drop_code0 + (uint32_t)t,
                                                 video_id, sample_id, st);
        if (e != hipSuccess) return e;
        if (t > 0) {
            e = nn_bwd(dZ + (size_t)t * 4 * MH, 4 * H, W + (size_t)kw0 * 4 * H, 4 * H, sc.slab, H, M, H, 4 * H, sp.splits, MH, st);
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

}  // namespace

extern "C" {

size_t s2vt_lstm_recurrence_bwd_scratch_bytes(int32_t M, int32_t H)
{
    if (M <= 0 || H <= 0) return 0;
    Carver c(nullptr, 0);
    size_t imgf, exf, syncb;
    bwd_scratch_max(H, M, &imgf, &exf, &syncb);
    c.take<float>((size_t)kMaxSlabs * M * H); c.take<float>((size_t)M * H);
    c.take<float>(imgf); c.take<float>(exf); c.take<unsigned>(syncb / 4);
    return c.off;
}

int s2vt_lstm_recurrence_bwd(const float* W, int32_t kw0, const float* gates, const float* C_hist, const float* dext, int64_t dext_tstride,
                             int32_t ld_ext, int32_t dext_t0, float* dZ, int32_t M, int32_t H, int32_t T, float keep, uint64_t seed,
                             const int32_t* video_id, const int32_t* sample_id, uint32_t drop_code0, int32_t persistent, void* scratch,
                             size_t scratch_bytes, s2vt_stream stream)
{
    if (!W || !gates || !C_hist || !dZ || M <= 0 || H <= 0 || T < 0 || kw0 < 0 || persistent < -1 || persistent > 1 || !scratch) return S2VT_E_BADARG;
    if (dext && (ld_ext < H || dext_t0 < 0)) return S2VT_E_BADARG;
    if (!(keep > 0.0f) || (keep < 1.0f && (!video_id || !sample_id))) return S2VT_E_BADARG;
    if (reinterpret_cast<uintptr_t>(scratch) & 255u) return S2VT_E_ALIGN;
    if (chain_fault()) return S2VT_E_CHAIN_TIMEOUT;
    Carver c(scratch, scratch_bytes);
    size_t imgf, exf, syncb;
    bwd_scratch_max(H, M, &imgf, &exf, &syncb);
    BwdScratch sc;
    sc.slab = c.take<float>((size_t)kMaxSlabs * M * H); sc.dc = c.take<float>((size_t)M * H);
    sc.bimg = c.take<float>(imgf); sc.bex = c.take<float>(exf); sc.bsync = c.take<unsigned>(syncb / 4);
    if (!c.ok()) return S2VT_E_WORKSPACE;
    if (persistent == 1 && !bwd_chain_eligible(M, H)) return S2VT_E_BADARG;
    HIP_TRY(lstm_recurrence_bwd(W, kw0, gates, C_hist, dext, (size_t)dext_tstride, ld_ext, dext_t0, dZ, M, H, T, keep, seed, drop_code0, video_id,
                                sample_id, sc, persistent, S(stream)));
    return S2VT_OK;
}

size_t s2vt_train_workspace_bytes(const s2vt_dims* d, int32_t B, int32_t N)
{
    if (!dims_ok(d) || B <= 0 || N <= 0 || N % B) return 0;
    Carver c(nullptr, 0);
    return carve_train(c, d, B, N, nullptr);
}

int s2vt_teacher_forced_fwd(const s2vt_dims* d, const s2vt_params* p, const float* video, int32_t B, int32_t N,
                            const int32_t* caption, float keep, uint64_t seed, const int32_t* video_id,
                            const int32_t* sample_id, float* logits, void* workspace, size_t workspace_bytes,
                            s2vt_stream stream)
{
    return s2vt_teacher_forced_fwd_reuse(d, p, video, B, N, caption, keep, seed, video_id, sample_id, logits, workspace,
                                         workspace_bytes, nullptr, 0, 0, stream);
}

int s2vt_teacher_forced_fwd_reuse(const s2vt_dims* d, const s2vt_params* p, const float* video, int32_t B, int32_t N,
                                  const int32_t* caption, float keep, uint64_t seed, const int32_t* video_id,
                                  const int32_t* sample_id, float* logits, void* workspace, size_t workspace_bytes,
                                  const void* sampler_workspace, size_t sampler_workspace_bytes, int32_t sampler_rows,
                                  s2vt_stream stream)
{
    return s2vt_teacher_forced_fwd_steps(d, p, video, B, N, caption, d ? d->n_caption_lstm_step : 0, keep, seed, video_id, sample_id,
                                         logits, workspace, workspace_bytes, sampler_workspace, sampler_workspace_bytes, sampler_rows,
                                         stream);
}

int s2vt_teacher_forced_fwd_steps(const s2vt_dims* d, const s2vt_params* p, const float* video, int32_t B, int32_t N,
                                  const int32_t* caption, int32_t caption_steps, float keep, uint64_t seed,
                                  const int32_t* video_id, const int32_t* sample_id, float* logits, void* workspace,
                                  size_t workspace_bytes, const void* sampler_workspace, size_t sampler_workspace_bytes,
                                  int32_t sampler_rows, s2vt_stream stream)
{
    return s2vt_teacher_forced_fwd_live(d, p, video, B, N, caption, caption_steps, nullptr, 0, keep, seed, video_id, sample_id, logits,
                                        workspace, workspace_bytes, sampler_workspace, sampler_workspace_bytes, sampler_rows, stream);
}

int s2vt_teacher_forced_fwd_live(const s2vt_dims* d, const s2vt_params* p, const float* video, int32_t B, int32_t N,
                                 const int32_t* caption, int32_t caption_steps, const int32_t* live_rows, int32_t n_live, float keep,
                                 uint64_t seed, const int32_t* video_id, const int32_t* sample_id, float* logits, void* workspace,
                                 size_t workspace_bytes, const void* sampler_workspace, size_t sampler_workspace_bytes,
                                 int32_t sampler_rows, s2vt_stream stream)
{
    if ((live_rows == nullptr) != (n_live == 0) || n_live < 0) return S2VT_E_BADARG;
    if (!dims_ok(d) || !params_ok(p) || !video || !caption || !logits || !workspace || B <= 0 || N <= 0 || N % B)
        return S2VT_E_BADARG;
    if (!(keep > 0.0f) || (keep < 1.0f && (!video_id || !sample_id))) return S2VT_E_BADARG;
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) return S2VT_E_ALIGN;
    if (chain_fault()) return S2VT_E_CHAIN_TIMEOUT;
    if (caption_steps < 1 || caption_steps > d->n_caption_lstm_step) return S2VT_E_BADARG;
    if (live_rows && (n_live > caption_steps * N || ((d->lstm_dim | d->word_dim) & 3))) return S2VT_E_BADARG;     // (packed rows move as 16-byte pieces)
    // Tc / T below are the steps this call UNROLLS; every buffer is time-major, so a truncated unroll is the leading part
    // of the full one's layout (the workspace is carved by the model's dimensions whatever caption_steps says)
    const int H = d->lstm_dim, E = d->word_dim, V = d->n_words, Tv = d->n_video_lstm_step, Tc = caption_steps;
    const int T = Tv + Tc;
    Carver c(workspace, workspace_bytes);
    TrainWs w;
    carve_train(c, d, B, N, &w);
    if (!c.ok()) return S2VT_E_WORKSPACE;
    hipStream_t st = S(stream);

    HIP_TRY(launch_prep_caption(caption, w.prev, w.tgt, N, d->n_caption_lstm_step, st));
    hipLaunchKernelGGL(enc_index_kernel, dim3((B * Tv + 255) / 256), dim3(256), 0, st, w.encidx, B, Tv);
    HIP_TRY(hipGetLastError());
    const size_t NH = (size_t)N * H, BH = (size_t)B * H;
    // zero initial states (tf_s2vt.py:105-107): slot 0 of the state histories
    {
        ZeroList z;
        if (!sampler_workspace) { z.add(w.C1, BH * 4); z.add(w.H1, BH * 4); }      // (taken whole from the sampler pass otherwise)
        z.add(w.C2, NH * 4); z.add(w.H2, NH * 4);
        HIP_TRY(launch_zero_regions(z, st));
    }
    if (!sampler_workspace) {                     // (with the sampler's workspace the frame embedding is taken from there too, below)
        int rc = s2vt_frame_embed_fwd(d, p, video, B, w.emb, stream);
        if (rc != S2VT_OK) return rc;
    }

    NoiseIds none{nullptr, nullptr, 0};
    NoiseIds ids{video_id, sample_id, seed};
    // Every contraction below is the SAME ascending-k chain as concat([x, h]) @ W (tf_s2vt.py:119-143):
    // the rows of W that multiply non-recurrent inputs are consumed first, for all time steps in one
    // batched launch; the per-step launch then continues each chain from that partial with the
    // recurrent rows.
    // ---- LSTM1: input rows of W1 for the Tv frames (row j*Tv + t of emb), then the recurrence on B rows
    if (sampler_workspace) {
        // The sampler pass of this step already ran LSTM1 on these videos with these weights (its state never
        // sees a word or a dropout mask): take its state history and activated gates instead of recomputing.
        Carver sc(const_cast<void*>(sampler_workspace), sampler_workspace_bytes);
        SampleWs sw;
        carve_sample(sc, d, B, sampler_rows, &sw);
        if (!sc.ok() || sampler_rows <= 0) return S2VT_E_WORKSPACE;
        CopyList cl;
        // (the frame embedding the backward's dW1 product reads: the same product of the same operands in the sampler pass)
        if (cl.add(w.C1, sw.c1, (size_t)(T + 1) * BH * 4) && cl.add(w.H1, sw.h1, (size_t)(T + 1) * BH * 4) && cl.add(w.G1, sw.G1, (size_t)T * 4 * BH * 4) &&
            cl.add(w.emb, sw.emb, (size_t)B * Tv * E * 4)) {
            HIP_TRY(launch_copy_regions(cl, st));                                  // one library launch
        } else {                                                                   // (B * H not a multiple of 4: the runtime's copies)
            HIP_TRY(hipMemcpyAsync(w.emb, sw.emb, (size_t)B * Tv * E * 4, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(w.C1, sw.c1, (size_t)(T + 1) * BH * 4, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(w.H1, sw.h1, (size_t)(T + 1) * BH * 4, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(w.G1, sw.G1, (size_t)T * 4 * BH * 4, hipMemcpyDeviceToDevice, st));
        }
    } else {
        {
            ASeg sx = make_seg(w.emb, E, E, 0);
            HIP_TRY(store_call(&sx, 1, p->lstm1_W, 4 * H, nullptr, w.Xp1, 4 * H, B * Tv, 4 * H, 0, -1, st));
        }
        // tf_s2vt.py:119 (encode) / :140 (decode, zero padding input: only the recurrent rows remain)
        HIP_TRY(lstm_recurrence(p->lstm1_W, E, p->lstm1_b, w.Xp1, (size_t)4 * H, Tv * 4 * H, Tv, w.C1, w.H1, BH, w.G1, (size_t)4 * BH,
                                nullptr, 0, B, H, T, 1.0f, none, 0, w.chain_abuf, w.chain_sync, st));
    }
    // DropoutWrapper(LSTM1) output for the N sample rows (tf_s2vt.py:75; code = 256 + t)
    HIP_TRY(launch_expand_dropout(w.H1 + BH, w.O1, T, B, N, H, keep, seed, 256u, video_id, sample_id, st));
    // ---- LSTM2: rows of W2 for [out1 ; embed(prev word)] for all steps at once, written where the
    // step's activated gates will go (the step kernel reads its partial, then overwrites it)
    {
        ASeg se = make_seg(w.O1, H, H, 0);                                       // encode: the word slot is the zero padding (:122)
        HIP_TRY(store_call(&se, 1, p->lstm2_W, 4 * H, nullptr, w.G2, 4 * H, Tv * N, 4 * H, 0, -1, st));
        if (!live_rows) {
            ASeg sd[2] = {make_seg(w.O1 + (size_t)Tv * NH, H, H, 0), make_seg(p->Wemb, E, E, H, 0, w.prev)};   // decode (:143)
            HIP_TRY(store_call(sd, 2, p->lstm2_W, 4 * H, nullptr, w.G2 + (size_t)Tv * 4 * NH, 4 * H, Tc * N, 4 * H, 0, -1, st));
        } else {
            // live rows: the partials of the unmasked positions only (both operands gathered through the list), packed, then
            // scattered into the zeroed decode part -- a masked position's cell then steps from a zero partial: finite values
            // that feed nothing (its logits are not computed, its gradients are zeros)
            HIP_TRY(launch_gather_i32(w.prev, live_rows, n_live, w.prevp, st));
            ASeg sd[2] = {make_seg(w.O1 + (size_t)Tv * NH, H, H, 0, 0, live_rows), make_seg(p->Wemb, E, E, H, 0, w.prevp)};
            HIP_TRY(store_call(sd, 2, p->lstm2_W, 4 * H, nullptr, w.dZ2p, 4 * H, n_live, 4 * H, 0, -1, st));   // (dZ2p: free until the backward)
            float* const dec = w.G2 + (size_t)Tv * 4 * NH;
            ZeroList z;
            z.add(dec, (size_t)Tc * N * 4 * H * 4);
            // (recurrences that stop the rows behind their <eos> leave those rows' histories unwritten: G2 keeps the zeros of this
            //  fill there, and the cell states get zeros too, so that a DENSE backward recurrence over this workspace -- the per-step
            //  form under s2vt_chain_hold -- multiplies its zero gradients with finite values)
            if (live_rows && Tc <= 128 && chain_live_capable(N, H)) z.add(w.C2 + (size_t)(Tv + 1) * NH, (size_t)Tc * NH * 4);
            HIP_TRY(launch_zero_regions(z, st));
            HIP_TRY(launch_scatter_rows(w.dZ2p, 4 * H, live_rows, n_live, 4 * H, dec, 4 * H, st));
        }
    }
    // the recurrence continues each chain from its partial in G2[t] and overwrites it with the activated gates.  Live rows at
    // 257-384 rows: a row behind its <eos> stops stepping (rows by length, live rows per step -- its histories there are not
    // written and not read: every consumer below and in the backward pass gathers the live pairs)
    const bool rec_live = live_rows && Tc <= 128 && chain_live_capable(N, H);
    // (the kernel also VALIDATES the list -- per-row prefix, strictly ascending -- and raises the sticky fault otherwise, so it runs for
    //  every live pass it can take, whether or not the recurrences of this shape use the order)
    if (live_rows && Tc <= 128 && N <= 1024) HIP_TRY(launch_row_order(live_rows, n_live, N, Tv, Tc, w.perm, w.nlive, st));
    HIP_TRY(lstm_recurrence(p->lstm2_W, H + E, p->lstm2_b, w.G2, (size_t)4 * NH, 4 * H, T, w.C2, w.H2, NH, w.G2, (size_t)4 * NH,
                            w.O2, NH, N, H, T, keep, ids, 512u, w.chain_abuf, w.chain_sync, st, rec_live ? w.perm : nullptr,
                            rec_live ? w.nlive : nullptr));
    // vocab logits for all Tc steps at once (tf_s2vt.py:153): rows t*N + n -- or only the LIVE ones (row r of the output is
    // row live_rows[r] of the unroll: a masked position's logits feed nothing, its loss term and gradient are exact zeros)
    ASeg so = make_seg(w.O2 + (size_t)Tv * NH, H, H, 0, 0, live_rows);
    HIP_TRY(store_call(&so, 1, p->embed_word_W, V, p->embed_word_b, logits, V, live_rows ? n_live : Tc * N, V, 0, -1, st));
    return S2VT_OK;
}

int s2vt_caption_mask(const int32_t* ids, int32_t N, int32_t Tc, float* mask, int32_t* target_tm, float* mask_sum, float* mask_sum_copy, s2vt_stream stream)
{
    if (!ids || N <= 0 || Tc <= 0) return S2VT_E_BADARG;
    HIP_TRY(launch_caption_mask(ids, N, Tc, mask, target_tm, mask_sum, mask_sum_copy, S(stream)));
    return S2VT_OK;
}

int s2vt_pg_coef(const float* mask, const float* rewards, const float* baseline, float scale, int32_t N, int32_t Tc, float* coef_tm,
                 s2vt_stream stream)
{
    if (!mask || !coef_tm || N <= 0 || Tc <= 0) return S2VT_E_BADARG;
    HIP_TRY(launch_pg_coef(mask, rewards, baseline, scale, N, Tc, coef_tm, S(stream)));
    return S2VT_OK;
}

int s2vt_xe_prep(const float* mask, const int32_t* caption, int32_t N, int32_t Tc, float loss_weight, float n_global, int32_t q1, float* coef_tm,
                 int32_t* target_tm, float* mask_sum, s2vt_stream stream)
{
    if (!mask || !coef_tm || (target_tm && !caption) || N <= 0 || Tc <= 0 || Tc > 128 || !(n_global > 0.0f)) return S2VT_E_BADARG;
    HIP_TRY(launch_xe_prep(mask, caption, N, Tc, loss_weight, n_global, q1, coef_tm, target_tm, mask_sum, S(stream)));
    return S2VT_OK;
}

int s2vt_mixed_prep(const float* mask, const float* gt_mask, const float* rewards, const float* baseline, const int32_t* sampled,
                    const int32_t* gt_caption, int32_t Ns, int32_t B, int32_t Tc, double lambda_loss, float loss_weight, int32_t q1,
                    float smoothing, float n_global_b, float* coef_tm, float* smooth_tm, int32_t* caption_all, int32_t* target_tm, float* sums, s2vt_stream stream)
{
    if (!mask || !gt_mask || !rewards || !baseline || !sampled || !gt_caption || !coef_tm || !smooth_tm || !caption_all || !target_tm || !sums) return S2VT_E_BADARG;
    if (Ns <= 0 || B <= 0 || Tc <= 0 || Tc > 128 || !(n_global_b > 0.0f)) return S2VT_E_BADARG;
    HIP_TRY(launch_mixed_prep(mask, gt_mask, rewards, baseline, sampled, gt_caption, Ns, B, Tc, (float)(1.0 - lambda_loss), (float)lambda_loss,
                              loss_weight, q1, smoothing, n_global_b, coef_tm, smooth_tm, caption_all, target_tm, sums, S(stream)));
    return S2VT_OK;
}

int s2vt_mixed_loss(const float* coef, const float* nll, const int32_t* live_rows, int64_t R, int32_t N, int32_t Ns, float* out3, s2vt_stream stream)
{
    if (!coef || !nll || !out3 || R < 0 || R > 0x7fffffff || N <= 0 || Ns < 0 || Ns > N) return S2VT_E_BADARG;
    HIP_TRY(launch_mixed_loss(coef, nll, live_rows, (int)R, N, Ns, out3, S(stream)));
    return S2VT_OK;
}

int s2vt_gemm_nt_splitk(const float* A, int32_t lda, const float* Wt, int32_t ldw, float* C, int32_t ldc, int32_t M, int32_t N,
                        int32_t K, int32_t splits, int32_t tile_cfg, float* slabs, size_t slab_floats, s2vt_stream stream)
{
    if (!A || !Wt || !C || M < 0 || N <= 0 || K <= 0 || lda < K || ldw < K || ldc < N || splits < 0 || splits > 64) return S2VT_E_BADARG;
    if (M == 0) return S2VT_OK;
    hipStream_t st = S(stream);
    int s = splits == 0 ? dx_splits(M, N, K) : splits;
    if (s > 1 && (ldc != N || !slabs)) { if (splits) return S2VT_E_BADARG; s = 1; }
    if (s <= 1) { HIP_TRY(nn_bwd(A, lda, Wt, ldw, C, ldc, M, N, K, 1, 0, st, tile_cfg)); return S2VT_OK; }
    const size_t stride = (size_t)M * N;
    const int kper = ((K + s - 1) / s + BK - 1) / BK * BK, nslab = (K + kper - 1) / kper;     // what nn_bwd makes of `s`
    if ((size_t)nslab * stride > slab_floats) return S2VT_E_WORKSPACE;
    HIP_TRY(nn_bwd(A, lda, Wt, ldw, slabs, N, M, N, K, s, stride, st, tile_cfg >= 0 ? tile_cfg : kSlabTileCfg));
    HIP_TRY(launch_sum_slabs(C, slabs, nslab, stride, stride, st));
    return S2VT_OK;
}

int s2vt_step_scalars(const float* coef, const float* nll, int64_t R, const float* mask_sum_local, const float* mask_sum_global, float* loss,
                      float* gscale, float* sumsq, s2vt_stream stream)
{
    if (R < 0 || ((coef == nullptr) != (nll == nullptr))) return S2VT_E_BADARG;
    HIP_TRY(launch_step_scalars(coef, nll, R, mask_sum_local, mask_sum_global, loss, gscale, sumsq, S(stream)));
    return S2VT_OK;
}

int s2vt_softmax_nll_fwd_bwd(float* logits, int32_t ld, int32_t R, int32_t V, const int32_t* target, const float* coef,
                             float smoothing, float* nll, float* lp_target, s2vt_stream stream)
{
    if (!logits || !target || !coef || R < 0 || V <= 0 || ld < V) return S2VT_E_BADARG;
    HIP_TRY(launch_softmax_nll(logits, ld, R, V, target, coef, smoothing, nll, lp_target, S(stream)));
    return S2VT_OK;
}

int s2vt_softmax_unshifted_argmax(const float* logits, int32_t ld, int32_t R, int32_t V, int32_t* ids, float* probs,
                                  s2vt_stream stream)
{
    if (!logits || !ids || R < 0 || V <= 0 || ld < V) return S2VT_E_BADARG;
    HIP_TRY(launch_softmax_unshifted_argmax(logits, ld, R, V, ids, probs, S(stream)));
    return S2VT_OK;
}

int s2vt_softmax_nll_fwd_bwd_rows(float* logits, int32_t ld, int32_t R, int32_t V, const int32_t* target, const float* coef,
                                  const float* smoothing_rows, float* nll, float* lp_target, s2vt_stream stream)
{
    if (!logits || !target || !coef || !smoothing_rows || R < 0 || V <= 0 || ld < V) return S2VT_E_BADARG;
    HIP_TRY(launch_softmax_nll(logits, ld, R, V, target, coef, 0.0f, nll, lp_target, S(stream), smoothing_rows));
    return S2VT_OK;
}

int s2vt_bptt_bwd(const s2vt_dims* d, const s2vt_params* p, const s2vt_params* grads, const float* video, int32_t B,
                  int32_t N, const float* dlogits, float keep, uint64_t seed, const int32_t* video_id,
                  const int32_t* sample_id, void* workspace, size_t workspace_bytes, s2vt_stream stream)
{
    return s2vt_bptt_bwd_phase(d, p, grads, video, B, N, dlogits, keep, seed, video_id, sample_id, workspace, workspace_bytes, 0,
                               stream);
}

int s2vt_bptt_bwd_phase(const s2vt_dims* d, const s2vt_params* p, const s2vt_params* grads, const float* video, int32_t B,
                        int32_t N, const float* dlogits, float keep, uint64_t seed, const int32_t* video_id,
                        const int32_t* sample_id, void* workspace, size_t workspace_bytes, int32_t phase, s2vt_stream stream)
{
    return s2vt_bptt_bwd_steps(d, p, grads, video, B, N, dlogits, d ? d->n_caption_lstm_step : 0, keep, seed, video_id, sample_id,
                               workspace, workspace_bytes, phase, stream);
}

int s2vt_bptt_bwd_steps(const s2vt_dims* d, const s2vt_params* p, const s2vt_params* grads, const float* video, int32_t B,
                        int32_t N, const float* dlogits, int32_t caption_steps, float keep, uint64_t seed,
                        const int32_t* video_id, const int32_t* sample_id, void* workspace, size_t workspace_bytes, int32_t phase,
                        s2vt_stream stream)
{
    return s2vt_bptt_bwd_live(d, p, grads, video, B, N, dlogits, caption_steps, nullptr, 0, keep, seed, video_id, sample_id, workspace,
                              workspace_bytes, phase, stream);
}

int s2vt_bptt_bwd_live(const s2vt_dims* d, const s2vt_params* p, const s2vt_params* grads, const float* video, int32_t B,
                       int32_t N, const float* dlogits, int32_t caption_steps, const int32_t* live_rows, int32_t n_live, float keep,
                       uint64_t seed, const int32_t* video_id, const int32_t* sample_id, void* workspace, size_t workspace_bytes,
                       int32_t phase, s2vt_stream stream)
{
    if ((live_rows == nullptr) != (n_live == 0) || n_live < 0) return S2VT_E_BADARG;
    if (phase < 0 || phase > 4) return S2VT_E_BADARG;
    if (!dims_ok(d) || !params_ok(p) || !params_ok(grads) || !video || !dlogits || !workspace || B <= 0 || N <= 0 || N % B)
        return S2VT_E_BADARG;
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) return S2VT_E_ALIGN;
    if (chain_fault()) return S2VT_E_CHAIN_TIMEOUT;
    if (caption_steps < 1 || caption_steps > d->n_caption_lstm_step) return S2VT_E_BADARG;
    if (live_rows && (n_live > caption_steps * N || ((d->lstm_dim | d->word_dim) & 3))) return S2VT_E_BADARG;
    // the steps the forward call unrolled (s2vt_teacher_forced_fwd_steps): later steps carry no gradient, so the
    // recurrences start from zero at step T - 1 and every contraction covers the leading T (Tc) steps only
    const int H = d->lstm_dim, E = d->word_dim, V = d->n_words, D = d->dim_image, Tv = d->n_video_lstm_step,
              Tc = caption_steps;
    const int T = Tv + Tc;
    Carver c(workspace, workspace_bytes);
    TrainWs w;
    carve_train(c, d, B, N, &w);
    if (!c.ok()) return S2VT_E_WORKSPACE;
    hipStream_t st = S(stream);
    const size_t NH = (size_t)N * H;

    // Phases, for data-parallel callers that start a slice's all-reduce as soon as its gradients are final:
    //   1 = the vocab projection (embed_word_W / _b final);  3 = LSTM2's recurrence + its weight gradients (lstm2_W / _b
    //   final);  4 = everything after (dX2, LSTM1, Wemb, frame embedding);  2 = 3 + 4;  0 = all.
    const bool do_vocab = phase == 0 || phase == 1, do_l2 = phase == 0 || phase == 2 || phase == 3,
               do_rest = phase == 0 || phase == 2 || phase == 4;
    SideStream& ss = side_stream();
    // gated overlap (mode 2): only beside recurrences that run as ONE-part persistent grids (<= 256 rows: room for a second wave per SIMD)
    const bool gate2 = ss.ok && ss.mode == 2 && phase == 0 && N <= 256 && bwd_chain_auto(N, H) && !(reinterpret_cast<uintptr_t>(p->lstm2_W) & 15);
    // (LSTM1's recurrence is gated only where LSTM2's is: at N > 256 rows -- the REINFORCE step's 320 -- LSTM2's three contractions are 0.6 ms
    //  each and lose more beside the 64-row recurrence than it gains: 12.26 -> 12.28 ms measured, profiles/r05_overlap_ab.jsonl)
    const bool gate1 = ss.ok && ss.mode == 2 && phase == 0 && N <= 256 && B <= 256 && bwd_chain_auto(B, H) && !(reinterpret_cast<uintptr_t>(p->lstm1_W) & 15);
    const bool side_on = ss.ok && phase == 0 && (ss.mode == 1 || gate1 || gate2);
    hipStream_t sd = side_on ? ss.s : st;                    // weight-gradient work that may run beside a recurrence (whole-pass calls only)
    // The side stream and its events are ONE per process: two host threads driving distinct workspaces must not interleave their
    // record / wait pairs (a wait enqueued after the OTHER thread's record of the same event would order this call's side work behind
    // the wrong point).  The lock covers this call's enqueueing only -- the launches themselves are asynchronous as ever.
    std::unique_lock<std::mutex> side_lk(side_stream_mutex(), std::defer_lock);
    if (side_on) side_lk.lock();
    ChainGate gate{ss.s, ss.ev[3], false};
    TnArgs dwout;                                            // (mode 2: the vocabulary projection's weight gradient is launched behind LSTM2's recurrence)
    bool dwout_deferred = false;
    if (do_vocab) {
        // transposed weight copy for the data-gradient product + the vocab projection
        hipStream_t sv = (phase == 0 && ss.ok && ss.mode == 1) ? sd : st;     // (phase 1: its gradients must be final on the caller's stream)
        if (sv != st) HIP_TRY(fork_to(st, sv, ss.ev[0]));
        // rows of the vocabulary-side products: every unrolled (step, row) pair, or the LIVE ones only (dlogits is then
        // [n_live, V], row r belonging to row live_rows[r] of the unroll; the masked rows' dlogits are exact zeros in the full
        // form, so leaving them out of the reductions changes nothing and their dO2 rows are the zeros written below)
        const int R = live_rows ? n_live : Tc * N;
        TnArgs a{w.O2 + (size_t)Tv * NH, live_rows, H, dlogits, V, grads->embed_word_W, V, R, H, V, 1};
        a.gather_rows = live_rows ? Tc * N : 0;
        a.colsum = grads->embed_word_b;                     // the bias gradient rides in the same pass over dlogits
        if (gate2 && do_l2) { dwout = a; dwout_deferred = true; }
        else HIP_TRY(launch_gemm_tn(a, sv));
        float* const dO2t = live_rows ? w.dO2p : w.dO2;     // where the product lands: packed rows are scattered afterwards
        int s2 = w.dO2s ? do2_splits(R, H, V) : 1;
        if ((size_t)s2 * R > w.dO2s_rows) s2 = 1;            // (a live-row count between two step counts the carve did not see)
        if (s2 > 1) {
            const size_t stride = (size_t)R * H;
            HIP_TRY(nn_bwd(dlogits, V, p->embed_word_W, V, w.dO2s, H, R, H, V, s2, stride, st, kSlabTileCfg));
            const int kper = ((V + s2 - 1) / s2 + BK - 1) / BK * BK;       // what nn_bwd made of `splits`
            HIP_TRY(launch_sum_slabs(dO2t, w.dO2s, (V + kper - 1) / kper, stride, stride, st));
        } else {
            HIP_TRY(nn_bwd(dlogits, V, p->embed_word_W, V, dO2t, H, R, H, V, 1, 0, st));
        }
        if (live_rows) {
            ZeroList z;
            z.add(w.dO2, (size_t)Tc * N * H * 4);
            HIP_TRY(launch_zero_regions(z, st));
            HIP_TRY(launch_scatter_rows(w.dO2p, H, live_rows, R, H, w.dO2, H, st));
        }
    }
    bool l2_deferred = false;
    std::function<int()> l2_grads_fn;
    if (do_l2) {
    // ---- LSTM2 back through time (one persistent launch up to 128 rows: chain.hip)
    {
        BwdScratch sc{w.slab, w.dc, w.bimg, w.bex, w.bsync};
        // (live rows: the forward pass of this workspace stopped the rows behind their <eos>, see there -- the same order here)
        const bool rec_live = live_rows && Tc <= 128 && chain_live_capable(N, H);
        if (live_rows && Tc <= 128 && N <= 1024) HIP_TRY(launch_row_order(live_rows, n_live, N, Tv, Tc, w.perm, w.nlive, st));
        if (dwout_deferred) chain_gate_arm(&gate);
        const hipError_t re = lstm_recurrence_bwd(p->lstm2_W, H + E, w.G2, w.C2, w.dO2, NH, H, Tv, w.dZ2, N, H, T, keep, seed, 512u, video_id, sample_id, sc, -1, st,
                                                  rec_live ? w.perm : nullptr, rec_live ? w.nlive : nullptr);
        chain_gate_arm(nullptr);
        HIP_TRY(re);
        if (dwout_deferred) {
            // dWout = O2^T dlogits beside the recurrence (reads dlogits and O2, writes embed_word_W's gradient: nothing the recurrence touches);
            // the gate has put the side stream behind "every workgroup of the grid resident" -- or, when the recurrence took its per-step
            // form after all (a hold, a fault), the side stream simply follows the caller's
            if (!gate.fired) HIP_TRY(fork_to(st, sd, ss.ev[3]));
            HIP_TRY(launch_gemm_tn(dwout, sd));
        }
    }
    // Live rows: the packed copies of the live dZ2 rows / previous words are read on BOTH streams (weight gradients on the side
    // stream, dX2 and the embedding scatter on the caller's), so they are made on the caller's stream, ahead of the fork
    if (live_rows) {
        HIP_TRY(launch_gather_rows(w.dZ2 + (size_t)Tv * 4 * NH, 4 * H, live_rows, n_live, 4 * H, w.dZ2p, 4 * H, st));
        HIP_TRY(launch_gather_i32(w.prev, live_rows, n_live, w.prevp, st));
    }
    // dZ2 is complete: LSTM2's weight gradients go to the side stream, beside dX2 and LSTM1's recurrence (mode 2: beside LSTM1's recurrence
    // only -- launched from the lambda below once that grid is resident; dX2 keeps the chip to itself)
    if (sd != st) HIP_TRY(fork_to(st, sd, ss.ev[1]));
    auto l2_weight_grads = [&]() -> int {
    if (!live_rows) {
        TnArgs a{w.O1, nullptr, H, w.dZ2, 4 * H, grads->lstm2_W, 4 * H, T * N, H, 4 * H, 1};
        HIP_TRY(launch_gemm_tn(a, sd));
        TnArgs b{p->Wemb, w.prev, E, w.dZ2 + (size_t)Tv * 4 * NH, 4 * H, grads->lstm2_W + (size_t)H * 4 * H, 4 * H, Tc * N, E,
                 4 * H, 1};
        b.gather_rows = V;                                  // Wemb [V, E]
        HIP_TRY(launch_gemm_tn(b, sd));
        TnArgs e{w.H2, nullptr, H, w.dZ2, 4 * H, grads->lstm2_W + (size_t)(H + E) * 4 * H, 4 * H, T * N, H, 4 * H, 1};
        e.colsum = grads->lstm2_b;
        HIP_TRY(launch_gemm_tn(e, sd));
    } else {
        // Live rows: behind its first <eos> a row's dZ2 is an exact zero at every later step (zero upstream gradient, zero carried
        // dh / dc), so the weight gradients are reduced over the Tv encode steps (every row) plus the LIVE decode rows -- their
        // dZ2 rows as a packed copy, the matching activation rows gathered through the same list.
        const int R = n_live;
        TnArgs a0{w.O1, nullptr, H, w.dZ2, 4 * H, grads->lstm2_W, 4 * H, Tv * N, H, 4 * H, 1};
        HIP_TRY(launch_gemm_tn(a0, sd));
        TnArgs a1{w.O1 + (size_t)Tv * NH, live_rows, H, w.dZ2p, 4 * H, grads->lstm2_W, 4 * H, R, H, 4 * H, 1};
        a1.gather_rows = Tc * N;
        HIP_TRY(launch_gemm_tn(a1, sd));
        TnArgs b{p->Wemb, w.prevp, E, w.dZ2p, 4 * H, grads->lstm2_W + (size_t)H * 4 * H, 4 * H, R, E, 4 * H, 1};
        b.gather_rows = V;                                  // Wemb [V, E]
        HIP_TRY(launch_gemm_tn(b, sd));
        TnArgs e0{w.H2, nullptr, H, w.dZ2, 4 * H, grads->lstm2_W + (size_t)(H + E) * 4 * H, 4 * H, Tv * N, H, 4 * H, 1};
        e0.colsum = grads->lstm2_b;
        HIP_TRY(launch_gemm_tn(e0, sd));
        TnArgs e1{w.H2 + (size_t)Tv * NH, live_rows, H, w.dZ2p, 4 * H, grads->lstm2_W + (size_t)(H + E) * 4 * H, 4 * H, R, H, 4 * H, 1};
        e1.gather_rows = Tc * N;
        e1.colsum = grads->lstm2_b;
        HIP_TRY(launch_gemm_tn(e1, sd));
    }
    return S2VT_OK;
    };
    l2_deferred = gate1 && do_rest && sd != st;
    if (!l2_deferred) { const int rc = l2_weight_grads(); if (rc != S2VT_OK) return rc; }
    else l2_grads_fn = l2_weight_grads;
    }
    if (!do_rest) return S2VT_OK;
    // d[out1 ; embed] for every step at once -- with live rows: the encode steps, then the live decode rows (from the packed dZ2
    // of the LSTM2 phase), scattered into the zeroed decode part
    if (!live_rows) {
        HIP_TRY(nn_bwd_slabs(w.dZ2, 4 * H, p->lstm2_W, 4 * H, w.dX2, H + E, T * N, H + E, 4 * H, w.dXs, st, w.dXs_floats));
    } else {
        HIP_TRY(nn_bwd_slabs(w.dZ2, 4 * H, p->lstm2_W, 4 * H, w.dX2, H + E, Tv * N, H + E, 4 * H, w.dXs, st, w.dXs_floats));
        HIP_TRY(nn_bwd_slabs(w.dZ2p, 4 * H, p->lstm2_W, 4 * H, w.dX2p, H + E, n_live, H + E, 4 * H, w.dXs, st, w.dXs_floats));
        float* const dec = w.dX2 + (size_t)Tv * N * (H + E);
        ZeroList z;
        z.add(dec, (size_t)Tc * N * (H + E) * 4);
        HIP_TRY(launch_zero_regions(z, st));
        HIP_TRY(launch_scatter_rows(w.dX2p, H + E, live_rows, n_live, H + E, dec, H + E, st));
    }
    // ---- LSTM1 back through time, on the B per-video rows: the gradient w.r.t. its dropped output is
    // first reduced over the rep sample rows of each video (with their dropout masks)
    const size_t BH = (size_t)B * H;
    HIP_TRY(launch_reduce_dropout(w.dX2, H + E, w.dH1, T, B, N, H, keep, seed, 256u, video_id, sample_id, st));
    {
        BwdScratch sc{w.slab, w.dc, w.bimg, w.bex, w.bsync};
        if (l2_deferred) chain_gate_arm(&gate);
        const hipError_t re = lstm_recurrence_bwd(p->lstm1_W, E, w.G1, w.C1, w.dH1, BH, H, 0, w.dZ1, B, H, T, 1.0f, seed, 0u, nullptr, nullptr, sc, -1, st);
        chain_gate_arm(nullptr);
        HIP_TRY(re);
        if (l2_deferred) {                                   // LSTM2's three weight-gradient contractions, beside LSTM1's recurrence (they read dZ2: final since ev[1])
            const int rc = l2_grads_fn();
            if (rc != S2VT_OK) return rc;
        }
    }
    HIP_TRY(nn_bwd_slabs(w.dZ1, 4 * H, p->lstm1_W, 4 * H, w.dX1, E, Tv * B, E, 4 * H, w.dXs, st, w.dXs_floats));

    // ---- remaining weight gradients: one contraction over all unrolled steps per weight block
    {
        TnArgs f{w.emb, w.encidx, E, w.dZ1, 4 * H, grads->lstm1_W, 4 * H, Tv * B, E, 4 * H, 1};
        f.gather_rows = Tv * B;
        HIP_TRY(launch_gemm_tn(f, st));
        TnArgs g{w.H1, nullptr, H, w.dZ1, 4 * H, grads->lstm1_W + (size_t)E * 4 * H, 4 * H, T * B, H, 4 * H, 1};
        g.colsum = grads->lstm1_b;
        HIP_TRY(launch_gemm_tn(g, st));
        // embedding rows (gradient of tf.nn.embedding_lookup): scatter-add of the embed slice of dX2
        if (live_rows) HIP_TRY(launch_scatter_add_rows(w.dX2p + H, H + E, w.prevp, n_live, E, grads->Wemb, E, st));
        else HIP_TRY(launch_scatter_add_rows(w.dX2 + (size_t)Tv * N * (H + E) + H, H + E, w.prev, Tc * N, E, grads->Wemb, E, st));
        // frame embedding
        TnArgs h{video, w.encidx, D, w.dX1, E, grads->encode_image_W, E, Tv * B, D, E, 1};
        h.gather_rows = Tv * B;
        h.colsum = grads->encode_image_b;
        HIP_TRY(launch_gemm_tn(h, st));
    }
    if (sd != st) HIP_TRY(fork_to(sd, st, ss.ev[2]));        // join: the caller's stream waits for the side stream's gradients
    return S2VT_OK;
}

int s2vt_bptt_dvideo(const s2vt_dims* d, const s2vt_params* p, int32_t B, int32_t N, float* d_video, void* workspace,
                      size_t workspace_bytes, s2vt_stream stream)
{
    if (!dims_ok(d) || !p || !p->encode_image_W || !d_video || !workspace || B <= 0 || N <= 0 || N % B) return S2VT_E_BADARG;
    Carver c(workspace, workspace_bytes);
    TrainWs w;
    carve_train(c, d, B, N, &w);
    if (!c.ok()) return S2VT_E_WORKSPACE;
    hipStream_t st = S(stream);
    const int E = d->word_dim, D = d->dim_image, Tv = d->n_video_lstm_step;
    // d_video[j*Tv + t, :] = dX1[t*B + j, :] @ encode_image_W^T   (dX1 is what s2vt_bptt_bwd left, time-major)
    hipLaunchKernelGGL(enc_index_kernel, dim3((B * Tv + 255) / 256), dim3(256), 0, st, w.decidx, Tv, B);
    HIP_TRY(hipGetLastError());
    GemmArgs a;
    std::memset(&a, 0, sizeof(a));
    a.seg[0] = make_seg(w.dX1, E, E, 0, 0, w.decidx);
    a.nseg = 1;
    a.W = p->encode_image_W; a.ldw = E; a.M = B * Tv; a.N = D; a.C = d_video; a.ldc = D;       // encode_image_W [D][E] is W^T as it lies
    HIP_TRY(launch_gemm(a, EPI_STORE_NT, -1, st));
    return S2VT_OK;
}

int s2vt_embed_scatter_add(const float* dE, int32_t ld, const int32_t* idx, int32_t R, int32_t E, float* dWemb,
                           s2vt_stream stream)
{
    if (!dE || !idx || !dWemb || R < 0 || E <= 0 || ld < E) return S2VT_E_BADARG;
    HIP_TRY(launch_scatter_add_rows(dE, ld, idx, R, E, dWemb, E, S(stream)));
    return S2VT_OK;
}

int s2vt_grad_finalize(float* g, const float* theta, int64_t n, const float* gscale, float weight_decay, float* sumsq,
                       s2vt_stream stream)
{
    if (!g || !sumsq || n < 0 || (weight_decay != 0.0f && !theta)) return S2VT_E_BADARG;
    HIP_TRY(launch_grad_finalize(g, theta, n, gscale, weight_decay, sumsq, S(stream)));
    return S2VT_OK;
}

int s2vt_adam_tf(float* theta, const float* g, float* m, float* v, int64_t n, const float* sumsq, float clip_norm,
                 float lr, int64_t step, float beta1, float beta2, float eps, s2vt_stream stream)
{
    return s2vt_adam_tf_guarded(theta, g, m, v, n, sumsq, clip_norm, lr, step, beta1, beta2, eps, nullptr, stream);
}

int s2vt_adam_tf_guarded(float* theta, const float* g, float* m, float* v, int64_t n, const float* sumsq, float clip_norm,
                         float lr, int64_t step, float beta1, float beta2, float eps, int32_t* applied_step, s2vt_stream stream)
{
    if (!theta || !g || !m || !v || n < 0 || step < 1) return S2VT_E_BADARG;
    if (chain_fault()) return S2VT_E_CHAIN_TIMEOUT;        // (a fault raised after this check is caught by the kernel itself)
    // lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t), epsilon outside the bias correction
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)step)) / (1.0 - pow((double)beta1, (double)step));
    HIP_TRY(launch_adam_tf(theta, g, m, v, n, sumsq, clip_norm, (float)lr_t, beta1, beta2, eps, S(stream), chain_fault_word(), applied_step,
                           (int32_t)step));
    return S2VT_OK;
}

}  // extern "C"
