// api.hip -- extern "C" boundary of libs2vt_hip.so (declared in include/s2vt.h) and the on-device
// drivers of the sampler loops.  Everything here is host code + a few trivial helper kernels; the
// contraction kernels live in gemm_mfma.h / fwd.hip.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "api_util.h"

namespace s2vt_api {
std::atomic<int> g_last_hip{0};
}
using namespace s2vt_api;

namespace {

__global__ void math_eval_kernel(int fn, const float* x, float* y, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    y[i] = fn == 0 ? dm_expf(v) : fn == 1 ? dm_logf(v) : fn == 2 ? dm_tanhf(v) : dm_sigmoidf(v);
}

__global__ void gumbel_eval_kernel(uint32_t lo, uint32_t hi, uint32_t video, uint32_t sample, uint32_t step, float* out,
                                   int V)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < V) out[n] = gumbel_at(lo, hi, video, sample, step, (uint32_t)n);
}

__global__ void fill_i32_kernel(int32_t* p, int32_t v, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// row ids of the sampler: row r = k*B + j -> video video_base + j, sample k (greedy block: -1)
__global__ void sampler_rows_kernel(int32_t* vid, int32_t* sid, int B, int K, int R, int video_base)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    vid[i] = video_base + i % B;
    sid[i] = (i / B) < K ? i / B : -1;
}

// packed [T][R] (entries `stride` words apart) -> ids [R][T].  A word that was never written (stop-at-<eos> mode: the row had
// left the loop) reads as <eos> = 0.
__global__ void unpack_ids_kernel(const unsigned long long* packed, int32_t* ids, int R, int T, int stride)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * T) return;
    const int m = i / T, t = i % T;
    const unsigned long long w = packed[((size_t)t * R + m) * stride];
    ids[i] = w ? (int32_t)(~(uint32_t)w) : 0;
}

// stop-at-<eos> mode: the rows still sampling at step t from those of step t - 1 and the words they just picked (a row leaves
// once it has picked <eos> = 0; order preserved).  One workgroup; step 0: every row.
__global__ __launch_bounds__(256) void live_rows_kernel(const unsigned long long* picked, int stride, const int32_t* prev, const int32_t* nprev,
                                                        int32_t* next, int32_t* nnext, int R)
{
    __shared__ int cnt[256];
    __shared__ int base;
    const int tid = threadIdx.x;
    if (!picked) {                                              // step 0
        for (int i = tid; i < R; i += 256) next[i] = i;
        if (tid == 0) *nnext = R;
        return;
    }
    const int n = *nprev;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int i = c0 + tid;
        int row = -1, alive = 0;
        if (i < n) {
            row = prev[i];
            alive = (uint32_t)(~(uint32_t)picked[(size_t)row * stride]) != 0u;
        }
        cnt[tid] = alive;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {                     // inclusive scan
            const int v = tid >= o ? cnt[tid - o] : 0;
            __syncthreads();
            cnt[tid] += v;
            __syncthreads();
        }
        if (alive) next[base + cnt[tid] - 1] = row;
        __syncthreads();
        if (tid == 255) base += cnt[255];
        __syncthreads();
    }
    if (tid == 0) *nnext = base;
}

}  // namespace

extern "C" {

int s2vt_version(void) { return 100; }

int s2vt_zero_regions(void* const* ptrs, const size_t* bytes, int32_t count, s2vt_stream stream)
{
    if (count < 0 || count > 8 || (count && (!ptrs || !bytes))) return S2VT_E_BADARG;
    ZeroList z;
    for (int i = 0; i < count; ++i) {
        if (bytes[i] & 3u) return S2VT_E_BADARG;
        if (ptrs[i] && (reinterpret_cast<uintptr_t>(ptrs[i]) & 3u)) return S2VT_E_ALIGN;
        z.add(ptrs[i], bytes[i]);
    }
    HIP_TRY(launch_zero_regions(z, S(stream)));
    return S2VT_OK;
}
int s2vt_last_hip_error(void) { return g_last_hip.load(); }

const char* s2vt_error_string(int code)
{
    switch (code) {
        case S2VT_OK: return "ok";
        case S2VT_E_BADARG: return "bad argument";
        case S2VT_E_ALIGN: return "workspace not 256-byte aligned";
        case S2VT_E_WORKSPACE: return "workspace too small";
        case S2VT_E_HIP: return "HIP error (see s2vt_last_hip_error)";
        case S2VT_E_CHAIN_TIMEOUT: return "a persistent recurrence timed out (GPU shared with another persistent kernel?): results since are suspect, updates were skipped; s2vt_chain_ack() and repeat";
        default: return "unknown error";
    }
}

int s2vt_prof_enable(int on)
{
    prof_enable(on != 0);
    return S2VT_OK;
}

int s2vt_prof_filter(int kernel_class, int tile_cfg)
{
    prof_filter(kernel_class, tile_cfg);
    return S2VT_OK;
}

int s2vt_prof_collect(s2vt_prof_row* rows, int max_rows)
{
    if (!rows || max_rows <= 0) return S2VT_E_BADARG;
    ProfRow tmp[64];
    const int n = prof_collect(tmp, max_rows < 64 ? max_rows : 64);
    for (int i = 0; i < n; ++i) {
        rows[i].kernel_class = tmp[i].cls; rows[i].tile_cfg = tmp[i].cfg; rows[i].launches = tmp[i].launches;
        rows[i].total_ms = tmp[i].ms; rows[i].total_flops = tmp[i].flops;
        std::strncpy(rows[i].name, tmp[i].name, sizeof(rows[i].name) - 1);
        rows[i].name[sizeof(rows[i].name) - 1] = 0;
    }
    return n;
}

int s2vt_math_eval(int fn, const float* x, float* y, int64_t n, s2vt_stream stream)
{
    if (!x || !y || n < 0 || fn < 0 || fn > 3) return S2VT_E_BADARG;
    if (n == 0) return S2VT_OK;
    hipLaunchKernelGGL(math_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), fn, x, y, n);
    HIP_TRY(hipGetLastError());
    return S2VT_OK;
}

int s2vt_gemm_nt(const s2vt_operand* segs, int32_t nseg, const float* Wt, int32_t ldw, const float* bias, const float* Cinit,
                 int32_t ldcinit, float* C, int32_t ldc, int32_t M, int32_t N, int32_t act_tanh, int32_t tile_cfg,
                 s2vt_stream stream)
{
    if (!segs || nseg < 1 || nseg > 3 || !Wt || !C || M < 0 || N <= 0 || ldc < N) return S2VT_E_BADARG;
    if (Cinit && ldcinit < N) return S2VT_E_BADARG;
    ASeg a[3];
    int kw = 0;
    for (int i = 0; i < nseg; ++i) {
        if (segs[i].k < 0 || (segs[i].ptr && segs[i].ld < segs[i].k)) return S2VT_E_BADARG;
        seg_from_operand(a[i], &segs[i], kw);
        kw += segs[i].k;
    }
    if (ldw < kw) return S2VT_E_BADARG;
    if (M == 0) return S2VT_OK;
    HIP_TRY(store_call(a, nseg, Wt, ldw, bias, C, ldc, M, N, act_tanh ? 1 : 0, tile_cfg, S(stream), Cinit, ldcinit, true));
    return S2VT_OK;
}

int s2vt_gumbel_eval(uint64_t seed, int32_t video, int32_t sample, int32_t step, float* out, int32_t V, s2vt_stream stream)
{
    if (!out || V <= 0) return S2VT_E_BADARG;
    hipLaunchKernelGGL(gumbel_eval_kernel, dim3((V + 255) / 256), dim3(256), 0, S(stream), (uint32_t)seed,
                       (uint32_t)(seed >> 32), (uint32_t)video, (uint32_t)sample, (uint32_t)step, out, V);
    HIP_TRY(hipGetLastError());
    return S2VT_OK;
}

int s2vt_gemm(const s2vt_operand* segs, int32_t nseg, const float* W, int32_t ldw, const float* bias, const float* Cinit,
              int32_t ldcinit, float* C, int32_t ldc, int32_t M, int32_t N, int32_t act_tanh, int32_t tile_cfg,
              s2vt_stream stream)
{
    if (!segs || nseg < 1 || nseg > 3 || !W || !C || M < 0 || N <= 0 || ldw < N || ldc < N) return S2VT_E_BADARG;
    if (Cinit && ldcinit < N) return S2VT_E_BADARG;
    ASeg a[3];
    int kw = 0;
    for (int i = 0; i < nseg; ++i) {
        if (segs[i].k < 0 || (segs[i].ptr && segs[i].ld < segs[i].k)) return S2VT_E_BADARG;
        seg_from_operand(a[i], &segs[i], kw);
        kw += segs[i].k;
    }
    if (M == 0) return S2VT_OK;
    HIP_TRY(store_call(a, nseg, W, ldw, bias, C, ldc, M, N, act_tanh ? 1 : 0, tile_cfg, S(stream), Cinit, ldcinit));
    return S2VT_OK;
}

int s2vt_lstm_cell_fwd(const s2vt_operand* x0, const s2vt_operand* x1, const float* h_prev, const float* c_prev,
                       int32_t state_rowmod, const float* W, const float* b, float* c_new, float* h_new, float* out,
                       float* gates, int32_t M, int32_t H, float keep, uint64_t seed, const int32_t* video_id,
                       const int32_t* sample_id, uint32_t drop_code, int32_t tile_cfg, s2vt_stream stream)
{
    if (!h_prev || !c_prev || !W || !b || !c_new || !h_new || M < 0 || H <= 0) return S2VT_E_BADARG;
    if (keep < 1.0f && (!video_id || !sample_id || !out || !(keep > 0.0f))) return S2VT_E_BADARG;
    ASeg a[3];
    int kw = 0, n = 0;
    const s2vt_operand* xs[2] = {x0, x1};
    for (int i = 0; i < 2; ++i) {
        if (!xs[i]) continue;
        if (xs[i]->k < 0 || (xs[i]->ptr && xs[i]->ld < xs[i]->k)) return S2VT_E_BADARG;
        seg_from_operand(a[n++], xs[i], kw);
        kw += xs[i]->k;
    }
    a[n++] = make_seg(h_prev, H, H, kw, state_rowmod);
    if (M == 0) return S2VT_OK;
    NoiseIds ids{video_id, sample_id, seed};
    HIP_TRY(lstm_call(a, n, W, b, c_prev, state_rowmod, c_new, h_new, out, gates, M, H, keep, ids, drop_code, tile_cfg,
                      S(stream)));
    return S2VT_OK;
}

int s2vt_vocab_pick(const float* out2, int32_t ld, const float* W, const float* b, int32_t M, int32_t H, int32_t V,
                    const int32_t* video_id, const int32_t* sample_id, int32_t step, uint64_t seed,
                    unsigned long long* packed, int32_t* tokens_out, float* logits_out, int32_t tile_cfg,
                    s2vt_stream stream)
{
    if (!out2 || !W || !b || !video_id || !sample_id || !packed || M < 0 || H <= 0 || V <= 0 || ld < H)
        return S2VT_E_BADARG;
    if (M == 0) return S2VT_OK;
    NoiseIds ids{video_id, sample_id, seed};
    HIP_TRY(pick_call(out2, ld, W, b, M, H, V, ids, step, packed, logits_out, tile_cfg, S(stream)));
    if (tokens_out) {
        hipLaunchKernelGGL(unpack_ids_kernel, dim3((M + 255) / 256), dim3(256), 0, S(stream), packed, tokens_out, M, 1, 1);
        HIP_TRY(hipGetLastError());
    }
    return S2VT_OK;
}

int s2vt_frame_embed_fwd(const s2vt_dims* d, const s2vt_params* p, const float* video, int32_t B, float* emb,
                         s2vt_stream stream)
{
    if (!dims_ok(d) || !p || !p->encode_image_W || !p->encode_image_b || !video || !emb || B < 0) return S2VT_E_BADARG;
    if (B == 0) return S2VT_OK;
    ASeg a = make_seg(video, d->dim_image, d->dim_image, 0);
    HIP_TRY(store_call(&a, 1, p->encode_image_W, d->word_dim, p->encode_image_b, emb, d->word_dim,
                       B * d->n_video_lstm_step, d->word_dim, 0, -1, S(stream)));
    return S2VT_OK;
}

// ---------------------------------------------------------------------------------------------
// samplers
// ---------------------------------------------------------------------------------------------
}  // extern "C"

namespace s2vt_api {

// Workspace of one sampler pass.  Everything up to `split` depends on B only (the encode half); the rest on
// the number of decode rows R.
size_t carve_sample(Carver& c, const s2vt_dims* d, int B, int R, SampleWs* w)
{
    const size_t H = d->lstm_dim, E = d->word_dim, Tv = d->n_video_lstm_step, Tc = d->n_caption_lstm_step, T = Tv + Tc;
    SampleWs t;
    t.emb = c.take<float>((size_t)B * Tv * E);
    t.Xp1 = c.take<float>((size_t)B * Tv * 4 * H);
    t.c1 = c.take<float>((T + 1) * B * H); t.h1 = c.take<float>((T + 1) * B * H);     // LSTM1 state history, slot 0 = zeros
    t.G1 = c.take<float>(T * B * 4 * H);                                                // LSTM1 activated gates (reused by the update pass)
    t.P2 = c.take<float>(T * B * 4 * H);                                                // h1[t+1] @ W2[0:H] for every step
    t.c2e = c.take<float>((Tv + 1) * (size_t)B * H); t.h2e = c.take<float>((Tv + 1) * (size_t)B * H);
    for (int i = 0; i < 2; ++i) { t.c2[i] = c.take<float>((size_t)R * H); t.h2[i] = c.take<float>((size_t)R * H); }
    t.packed = c.take<unsigned long long>((size_t)Tc * R * kPickStride);
    t.vid = c.take<int32_t>(R); t.sid = c.take<int32_t>(R); t.bos = c.take<int32_t>(R);
    t.chain_sync = c.take<unsigned>(kChainSyncBytes / 4);
    t.chain_abuf = c.take<float>(chain_scratch_floats((int)H));
    t.wemb_p = t.w2_p = t.himg[0] = t.himg[1] = nullptr;
    // fragment-order operands of the persistent decode loop (decode_loop.hip: default at <= 64 rows since round 6; S2VT_DECLOOP=2 / S2VT_DEC4=1 also at
    // 257-384 rows, where both forms measured no gain) -- sized by the shape and the two opt-in switches, never by the device: every caller of the
    // size query sees the same carve.  (24.6 + 40 MB of packed operands at the bench dimensions: not taken at 384 rows unless asked for.)
    static const bool big_rows = [] {
        const char* a = getenv("S2VT_DECLOOP"); const char* b = getenv("S2VT_DEC4");
        return (a && atoi(a) >= 2) || (b && b[0] == '1');
    }();
    static const bool small_rows = [] { const char* a = getenv("S2VT_DECLOOP"); return !a || atoi(a) >= 1; }();
    if (((R <= 64 && small_rows) || (R > 256 && R <= 384 && big_rows)) && (H & 3) == 0 && H >= 132 &&
        (size_t)d->n_words * ((E + 15) / 16 * 16) * 4 < (1ull << 31)) {
        Dec4Geom q;
        decode4_geometry(R, (int)H, (int)E, &q);
        t.wemb_p = c.take<float>((size_t)d->n_words * q.erow);
        t.w2_p = c.take<float>((size_t)q.ncg * 4 * q.ngt * 256);
        for (int i = 0; i < 2; ++i) t.himg[i] = c.take<float>((size_t)q.img_tiles * q.hgp * 256);
    }
    t.live[0] = c.take<int32_t>(R); t.live[1] = c.take<int32_t>(R); t.nlive = c.take<int32_t>(Tc + 1);
    if (w) *w = t;
    return c.off;
}

hipError_t lstm_recurrence(const float* W, int kw0, const float* bias, const float* cinit, size_t cinit_tstride, int ldcinit,
                           int cinit_steps, float* C, float* Hh, size_t state_tstride, float* gates, size_t gates_tstride,
                           float* out, size_t out_tstride, int M, int H, int T, float keep, const NoiseIds& ids,
                           uint32_t drop_code0, float* chain_abuf, unsigned* chain_sync, hipStream_t st, const int32_t* perm, const int32_t* nlive)
{
    if (chain_abuf && chain_sync && chain_eligible(M, H) && chain_operands_ok(W, 4 * H, chain_abuf)) {    // (unaligned W: per-step launches)
        ChainArgs a;
        std::memset(&a, 0, sizeof(a));
        a.W = W; a.ldw = 4 * H; a.kw0 = kw0; a.bias = bias;
        a.cinit = cinit; a.cinit_tstride = cinit_tstride; a.ldcinit = ldcinit; a.cinit_steps = cinit_steps;
        a.h0 = Hh; a.c0 = C; a.C = C; a.Hh = Hh; a.state_tstride = state_tstride;
        a.gates = gates; a.gates_tstride = gates_tstride; a.out = out; a.out_tstride = out_tstride;
        a.M = M; a.H = H; a.T = T; a.keep = keep;
        a.seed_lo = (uint32_t)ids.seed; a.seed_hi = (uint32_t)(ids.seed >> 32); a.drop_code0 = drop_code0;
        a.video_id = ids.video_id; a.sample_id = ids.sample_id;
        a.abuf = chain_abuf; a.sync = chain_sync;
        a.perm = perm; a.nlive = nlive;                        // (forms that cannot skip rows run dense: the same results)
        return launch_lstm_chain(a, st);
    }
    for (int t = 0; t < T; ++t) {
        ASeg s1 = make_seg(Hh + (size_t)t * state_tstride, H, H, kw0);
        hipError_t e = lstm_call(&s1, 1, W, bias, C + (size_t)t * state_tstride, 0, C + (size_t)(t + 1) * state_tstride,
                                 Hh + (size_t)(t + 1) * state_tstride, out ? out + (size_t)t * out_tstride : nullptr,
                                 gates ? gates + (size_t)t * gates_tstride : nullptr, M, H, keep, ids, drop_code0 + (uint32_t)t, -1, st,
                                 (cinit && t < cinit_steps) ? cinit + (size_t)t * cinit_tstride : nullptr, ldcinit, 0);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// Encoding stage (tf_s2vt.py:97-122) plus everything of the decoding stage that does not depend on a
// sampled word: frame embedding, the whole LSTM1 trajectory, its products with the out1 rows of W2, LSTM2 over
// the Tv frames.  Leaves the encoder state in slot Tv of w.c2e / w.h2e.
int sample_encode(const s2vt_dims* d, const s2vt_params* p, const float* video, int B, const SampleWs& w, s2vt_stream stream)
{
    const int H = d->lstm_dim, E = d->word_dim, Tv = d->n_video_lstm_step, Tc = d->n_caption_lstm_step;
    hipStream_t st = S(stream);
    const int T = Tv + Tc;
    const size_t BH = (size_t)B * H;
    // zero initial states (tf_s2vt.py:105-107)
    {
        ZeroList z;
        z.add(w.c1, BH * 4); z.add(w.h1, BH * 4); z.add(w.c2e, BH * 4); z.add(w.h2e, BH * 4);
        HIP_TRY(launch_zero_regions(z, st));
    }
    int rc = s2vt_frame_embed_fwd(d, p, video, B, w.emb, stream);
    if (rc != S2VT_OK) return rc;

    NoiseIds none{nullptr, nullptr, 0};
    // Each cell product is the ascending-k chain of concat([x, h]) @ W (tf_s2vt.py:119-143).  The rows of W
    // that multiply inputs known before the step are consumed first, batched over time; the per-step
    // launch continues the chain from that partial with the rows whose inputs the step produces.
    // ---- LSTM1: sees the frames, then only the zero padding and its own state -- never a sampled word,
    // so its whole trajectory (Tv + Tc steps) is per VIDEO and runs first, on B rows.
    {
        ASeg sx = make_seg(w.emb, E, E, 0);
        HIP_TRY(store_call(&sx, 1, p->lstm1_W, 4 * H, nullptr, w.Xp1, 4 * H, B * Tv, 4 * H, 0, -1, st));
    }
    // (one persistent launch for the whole trajectory when B <= 64: chain.hip)
    HIP_TRY(lstm_recurrence(p->lstm1_W, E, p->lstm1_b, w.Xp1, (size_t)4 * H, Tv * 4 * H, Tv, w.c1, w.h1, BH, w.G1, (size_t)4 * BH,
                            nullptr, 0, B, H, T, 1.0f, none, 0, w.chain_abuf, w.chain_sync, st));
    // ---- the out1 rows of W2 for every step at once (M = T*B)
    {
        ASeg so = make_seg(w.h1 + BH, H, H, 0);
        HIP_TRY(store_call(&so, 1, p->lstm2_W, 4 * H, nullptr, w.P2, 4 * H, T * B, 4 * H, 0, -1, st));
    }
    // ---- LSTM2 encoding stage (tf_s2vt.py:122: word slot = zero padding), M = B: the chain continues from the out1 partial
    HIP_TRY(lstm_recurrence(p->lstm2_W, H + E, p->lstm2_b, w.P2, (size_t)4 * BH, 4 * H, Tv, w.c2e, w.h2e, BH, nullptr, 0, nullptr, 0, B, H, Tv,
                            1.0f, none, 0, w.chain_abuf, w.chain_sync, st));
    return S2VT_OK;
}

// side streams of the row-group decode loop (S2VT_SAMPLE_GROUPS), created once per process
struct GroupStreams {
    hipStream_t s[2] = {nullptr, nullptr};
    hipEvent_t fork = nullptr, join[2] = {nullptr, nullptr};
    bool ok = false;
};
GroupStreams& group_streams()
{
    static GroupStreams gs = [] {
        GroupStreams t;
        for (auto& x : t.s)
            if (hipStreamCreateWithFlags(&x, hipStreamNonBlocking) != hipSuccess) return t;
        if (hipEventCreateWithFlags(&t.fork, hipEventDisableTiming) != hipSuccess) return t;
        for (auto& e : t.join)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return t;
        t.ok = true;
        return t;
    }();
    return gs;
}

// Decoding stage (tf_s2vt.py:126-153 as specialised by the samplers): LSTM2 + vocab at M = R rows, K
// multinomial row blocks then (with_greedy) one argmax block; the R rows of a video share its out1
// partial (row % B).  Needs sample_encode's results in the same workspace.
int sample_decode(const s2vt_dims* d, const s2vt_params* p, int B, int K, int with_greedy, uint64_t seed, int video_base,
                  int32_t* ids_out, const SampleWs& w, s2vt_stream stream, int stop_at_eos)
{
    const int H = d->lstm_dim, E = d->word_dim, V = d->n_words, Tv = d->n_video_lstm_step, Tc = d->n_caption_lstm_step;
    const int R = (K + (with_greedy ? 1 : 0)) * B;
    hipStream_t st = S(stream);
    const size_t BH = (size_t)B * H;
    {
        ZeroList z;
        z.add(w.packed, (size_t)Tc * R * kPickStride * 8);
        HIP_TRY(launch_zero_regions(z, st));
    }
    hipLaunchKernelGGL(sampler_rows_kernel, dim3((R + 255) / 256), dim3(256), 0, st, w.vid, w.sid, B, K, R, video_base);
    hipLaunchKernelGGL(fill_i32_kernel, dim3((R + 255) / 256), dim3(256), 0, st, w.bos, 1, R);   // <bos> = 1
    HIP_TRY(hipGetLastError());
    NoiseIds none{nullptr, nullptr, 0};
    NoiseIds ids{w.vid, w.sid, seed};
    const size_t enc = (size_t)Tv * B * H;   // where sample_encode left the encoder state: slot Tv of the history
    int cur2 = 0;
    // 257-384 rows: the LSTM2 step runs on fragment-order operands packed once per call (decode4.hip); same chain, same bits
    const bool loop1 = !stop_at_eos && w.wemb_p && (B & 15) == 0 && decode_loop_eligible(R, H, E, V) && chain_operands_ok(p->embed_word_W, V, w.himg[0]);
    const bool dec4 = !stop_at_eos && (loop1 || (w.wemb_p && decode4_eligible(R, H, E)));
    Dec4Geom q4;
    if (dec4) {
        decode4_geometry(R, H, E, &q4);
        HIP_TRY(decode4_pack(p->Wemb, p->lstm2_W, V, H, E, q4, w.wemb_p, w.w2_p, st));
        HIP_TRY(decode4_state_to_image(w.h2e + enc, B, R, H, q4, w.himg[0], st));
        HIP_TRY(hipMemsetAsync(w.himg[1], 0, (size_t)q4.img_tiles * q4.hgp * 1024, st));
    }
    if (loop1) {
        // all Tc steps -- LSTM2, vocabulary logits, pick -- in one persistent launch (decode_loop.hip)
        DecLoopLaunch a;
        std::memset(&a, 0, sizeof(a));
        a.wemb_p = w.wemb_p; a.w2_p = w.w2_p; a.bias2 = p->lstm2_b;
        a.P2 = w.P2 + (size_t)Tv * 4 * BH; a.p2_tstride = (size_t)4 * BH; a.ldp2 = 4 * H; a.B = B;
        a.c0 = w.c2e + enc;
        a.himg0 = w.himg[0]; a.himg1 = w.himg[1];
        a.packed = w.packed; a.pick_stride = kPickStride;
        a.Wout = p->embed_word_W; a.ldwo = V; a.bout = p->embed_word_b;
        a.seed = seed; a.video_base = video_base; a.noise_rows = K * B;
        a.R = R; a.H = H; a.E = E; a.V = V; a.Tc = Tc;
        a.sync = w.chain_sync;
        HIP_TRY(launch_decode_loop(a, q4, st));
        hipLaunchKernelGGL(unpack_ids_kernel, dim3((R * Tc + 255) / 256), dim3(256), 0, st, w.packed, ids_out, R, Tc, kPickStride);
        HIP_TRY(hipGetLastError());
        return S2VT_OK;
    }
    // Row groups (opt-in, S2VT_SAMPLE_GROUPS=2|3; round-4 verdict item 2): rows never interact, so the R rows can be cut into groups of whole
    // sample blocks, each group running its own chain of {LSTM2 step, pick} launches on a stream of its own -- pick(t) of one group
    // beside LSTM2(t+1) of another.  Same chains, same noise ids per row: token ids bit-identical.  Measured: see there.
    static const int n_groups = [] { const char* e = getenv("S2VT_SAMPLE_GROUPS"); const int g = e ? atoi(e) : 1; return g < 1 ? 1 : (g > 3 ? 3 : g); }();
    static const int g_lcfg = [] { const char* e = getenv("S2VT_GROUP_LSTM_CFG"); return e ? atoi(e) : -1; }();     // dev knobs: tiles of the group launches
    static const int g_pcfg = [] { const char* e = getenv("S2VT_GROUP_PICK_CFG"); return e ? atoi(e) : -1; }();
    const int blocks = R / B;
    if (n_groups > 1 && !dec4 && !stop_at_eos && blocks >= n_groups) {
        GroupStreams& gs = group_streams();
        if (gs.ok) {
            HIP_TRY(hipEventRecord(gs.fork, st));
            int b0 = 0;
            for (int g = 0; g < n_groups; ++g) {
                const int nb = blocks / n_groups + (g < blocks % n_groups ? 1 : 0);
                const int r0 = b0 * B, Rg = nb * B;
                b0 += nb;
                hipStream_t sg = g == 0 ? st : gs.s[g - 1];
                if (g > 0) HIP_TRY(hipStreamWaitEvent(sg, gs.fork, 0));
                NoiseIds idg{w.vid + r0, w.sid + r0, seed};
                int cur = 0;
                for (int t = 0; t < Tc; ++t) {
                    const int nxt = cur ^ 1;
                    const float* h2p = t == 0 ? w.h2e + enc : w.h2[cur] + (size_t)r0 * H;
                    const float* c2p = t == 0 ? w.c2e + enc : w.c2[cur] + (size_t)r0 * H;
                    const int smod = t == 0 ? B : 0;
                    ASeg s2[2] = {t == 0 ? make_seg(p->Wemb, E, E, H, 0, w.bos)
                                         : make_seg(p->Wemb, E, E, H, 0, nullptr, w.packed + ((size_t)(t - 1) * R + r0) * kPickStride, kPickStride),
                                  make_seg(h2p, H, H, H + E, smod)};
                    HIP_TRY(lstm_call(s2, 2, p->lstm2_W, p->lstm2_b, c2p, smod, w.c2[nxt] + (size_t)r0 * H, w.h2[nxt] + (size_t)r0 * H, nullptr, nullptr,
                                      Rg, H, 1.0f, none, 0, g_lcfg, sg, w.P2 + (size_t)(Tv + t) * 4 * BH, 4 * H, B));
                    HIP_TRY(pick_call(w.h2[nxt] + (size_t)r0 * H, H, p->embed_word_W, p->embed_word_b, Rg, H, V, idg, t,
                                      w.packed + ((size_t)t * R + r0) * kPickStride, nullptr, g_pcfg, sg, kPickStride));
                    cur = nxt;
                }
                if (g > 0) {
                    HIP_TRY(hipEventRecord(gs.join[g - 1], sg));
                    HIP_TRY(hipStreamWaitEvent(st, gs.join[g - 1], 0));
                }
            }
            hipLaunchKernelGGL(unpack_ids_kernel, dim3((R * Tc + 255) / 256), dim3(256), 0, st, w.packed, ids_out, R, Tc, kPickStride);
            HIP_TRY(hipGetLastError());
            return S2VT_OK;
        }
    }
    for (int t = 0; t < Tc; ++t) {
        const int nxt2 = cur2 ^ 1;
        if (dec4) {
            Dec4Launch a;
            std::memset(&a, 0, sizeof(a));
            a.wemb_p = w.wemb_p; a.w2_p = w.w2_p; a.bias = p->lstm2_b;
            a.cinit = w.P2 + (size_t)(Tv + t) * 4 * BH; a.ldcinit = 4 * H; a.cinit_rowmod = B;
            a.tok = t == 0 ? nullptr : w.packed + (size_t)(t - 1) * R * kPickStride; a.tok_stride = kPickStride; a.tok_const = 1;     // <bos> = 1
            a.himg_in = w.himg[t & 1]; a.himg_out = w.himg[(t + 1) & 1];
            a.c_prev = t == 0 ? w.c2e + enc : w.c2[cur2]; a.cprev_rowmod = t == 0 ? B : 0;
            a.c_new = w.c2[nxt2]; a.h_new = w.h2[nxt2];
            a.R = R; a.H = H; a.E = E; a.V = V;
            HIP_TRY(launch_decode_lstm4(a, q4, st));
            HIP_TRY(pick_call(w.h2[nxt2], H, p->embed_word_W, p->embed_word_b, R, H, V, ids, t, w.packed + (size_t)t * R * kPickStride,
                              nullptr, -1, st, kPickStride));
            cur2 = nxt2;
            continue;
        }
        const float* h2p = t == 0 ? w.h2e + enc : w.h2[cur2];
        const float* c2p = t == 0 ? w.c2e + enc : w.c2[cur2];
        const int smod = t == 0 ? B : 0;
        // stop-at-<eos> mode: the launch covers the rows still sampling (compact index -> row through the live list, their number on
        // the device); a finished row's state stays where it is, nothing reads it again, its later words are never written (= <eos>)
        const int* omap = nullptr;
        const int* mdev = nullptr;
        int lcfg = -1, pcfg = -1;
        if (stop_at_eos) {
            // tiles for a launch whose live-row count only the device knows: small row tiles cost a few % while every row is live
            // and follow the count down afterwards (dev knobs: S2VT_EOS_LSTM_CFG / S2VT_EOS_PICK_CFG)
            // -- measured at R = 384, mean length 7 (tools/eos_sweep.sh): cell step on 32-row tiles 2.40 ms per sampler call against 2.79
            // with the tile the full row count would get (96 rows) and 3.32 for the loop that never stops; the pick's 64 x 96 tile stays
            static const int lk = [] { const char* e = getenv("S2VT_EOS_LSTM_CFG"); return e ? atoi(e) : 4; }();     // kLstm[4] = gw32x16u
            static const int pk = [] { const char* e = getenv("S2VT_EOS_PICK_CFG"); return e ? atoi(e) : -1; }();
            lcfg = R > 64 ? lk : -1; pcfg = pk;
            hipLaunchKernelGGL(live_rows_kernel, dim3(1), dim3(256), 0, st, t == 0 ? nullptr : w.packed + (size_t)(t - 1) * R * kPickStride, kPickStride,
                               w.live[(t + 1) & 1], w.nlive + (t > 0 ? t - 1 : 0), w.live[t & 1], w.nlive + t, R);
            HIP_TRY(hipGetLastError());
            omap = w.live[t & 1]; mdev = w.nlive + t;
        }
        ASeg s2[2] = {t == 0 ? make_seg(p->Wemb, E, E, H, 0, w.bos)
                             : make_seg(p->Wemb, E, E, H, 0, nullptr, w.packed + (size_t)(t - 1) * R * kPickStride, kPickStride),
                      make_seg(h2p, H, H, H + E, smod)};
        HIP_TRY(lstm_call(s2, 2, p->lstm2_W, p->lstm2_b, c2p, smod, w.c2[nxt2], w.h2[nxt2], nullptr, nullptr, R, H, 1.0f,
                          none, 0, lcfg, st, w.P2 + (size_t)(Tv + t) * 4 * BH, 4 * H, B, omap, mdev));
        HIP_TRY(pick_call(w.h2[nxt2], H, p->embed_word_W, p->embed_word_b, R, H, V, ids, t, w.packed + (size_t)t * R * kPickStride,
                          nullptr, pcfg, st, kPickStride, omap, mdev));
        cur2 = nxt2;
    }
    hipLaunchKernelGGL(unpack_ids_kernel, dim3((R * Tc + 255) / 256), dim3(256), 0, st, w.packed, ids_out, R, Tc, kPickStride);
    HIP_TRY(hipGetLastError());
    return S2VT_OK;
}

bool sampler_params_ok(const s2vt_params* p)
{
    return p && p->Wemb && p->encode_image_W && p->encode_image_b && p->lstm1_W && p->lstm1_b && p->lstm2_W && p->lstm2_b &&
           p->embed_word_W && p->embed_word_b;
}

}  // namespace s2vt_api

extern "C" {

int s2vt_build_flags(void)
{
    return 1;       // bit 0: the fragment-order decode kernels (decode4.hip, decode_loop.hip) are in this library (always, since round 6)
}

size_t s2vt_sample_workspace_bytes(const s2vt_dims* d, int32_t B, int32_t K, int32_t with_greedy)
{
    if (!dims_ok(d) || B <= 0 || K < 0) return 0;
    Carver c(nullptr, 0);
    return carve_sample(c, d, B, (K + (with_greedy ? 1 : 0)) * B, nullptr);
}

int s2vt_sample(const s2vt_dims* d, const s2vt_params* p, const float* video, int32_t B, int32_t K, int32_t with_greedy,
                uint64_t seed, int32_t video_base, int32_t* ids_out, void* workspace, size_t workspace_bytes,
                s2vt_stream stream)
{
    if (!dims_ok(d) || !sampler_params_ok(p) || !video || !ids_out || !workspace || B <= 0 || K < 0 || (K == 0 && !with_greedy))
        return S2VT_E_BADARG;
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) return S2VT_E_ALIGN;
    if (chain_fault()) return S2VT_E_CHAIN_TIMEOUT;
    const int R = (K + (with_greedy ? 1 : 0)) * B;
    Carver c(workspace, workspace_bytes);
    SampleWs w;
    carve_sample(c, d, B, R, &w);
    if (!c.ok()) return S2VT_E_WORKSPACE;
    int rc = sample_encode(d, p, video, B, w, stream);
    if (rc != S2VT_OK) return rc;
    return sample_decode(d, p, B, K, with_greedy, seed, video_base, ids_out, w, stream);
}

int s2vt_sample_ex(const s2vt_dims* d, const s2vt_params* p, const float* video, int32_t B, int32_t K, int32_t with_greedy,
                   uint64_t seed, int32_t video_base, int32_t flags, int32_t* ids_out, void* workspace, size_t workspace_bytes,
                   s2vt_stream stream)
{
    if (!dims_ok(d) || !sampler_params_ok(p) || !video || !ids_out || !workspace || B <= 0 || K < 0 || (K == 0 && !with_greedy) || (flags & ~1))
        return S2VT_E_BADARG;
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) return S2VT_E_ALIGN;
    if (chain_fault()) return S2VT_E_CHAIN_TIMEOUT;
    const int R = (K + (with_greedy ? 1 : 0)) * B;
    Carver c(workspace, workspace_bytes);
    SampleWs w;
    carve_sample(c, d, B, R, &w);
    if (!c.ok()) return S2VT_E_WORKSPACE;
    int rc = sample_encode(d, p, video, B, w, stream);
    if (rc != S2VT_OK) return rc;
    return sample_decode(d, p, B, K, with_greedy, seed, video_base, ids_out, w, stream, flags & S2VT_SAMPLE_STOP_AT_EOS);
}

}  // extern "C"
