// session.hip -- the remaining entry points of the boundary listed in SURVEY.md section 8(b): a session handle
// that owns its sampler workspace (create / destroy / encode / decode_greedy / decode_multinomial), the
// single-op backward calls (frame embedding, LSTM cell), the two loss flavours by name (label-smoothed XE,
// reward-scaled policy-gradient NLL), embedding gather, global-norm clip, weight (re)packing between the
// TF BasicLSTMCell layout and a gate-interleaved split layout, and a thin RCCL all-reduce for C/C++ hosts.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include "api_util.h"
#include "chain_common.h"

using namespace s2vt_api;

struct s2vt_handle {
    s2vt_dims dims;
    int32_t max_B, max_K;
    void* ws;
    size_t ws_bytes;
    int32_t enc_B;          // batch of the last s2vt_encode_fwd (0 = none)
};

namespace {

__global__ void pack_lstm_kernel(const float* W, int rows, int H, float* P, int to_packed)
{
    // TF layout: W[r, g*H + u]  <->  gate-interleaved: P[r, u*4 + g]   (g = i, j, f, o)
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * 4 * H) return;
    const int col = (int)(i % (4 * (size_t)H)), r = (int)(i / (4 * (size_t)H));
    const int g = col / H, u = col % H;
    const size_t a = (size_t)r * 4 * H + col, b = (size_t)r * 4 * H + (size_t)u * 4 + g;
    if (to_packed) P[b] = W[a];
    else P[a] = W[b];
}

__global__ void pg_coef_kernel(const float* adv, const float* mask, float* coef, int N, int Tc)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // time-major row t*N + n
    if (i >= N * Tc) return;
    const int t = i / N, n = i % N;
    coef[i] = adv[n] * mask[n * Tc + t];
}

__global__ void gather_rows_kernel(const float* W, int ldw, const int32_t* idx, int R, int E, float* out, int ldo)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* src = W + (size_t)idx[r] * ldw;
    for (int e = threadIdx.x & 63; e < E; e += 64) out[(size_t)r * ldo + e] = src[e];
}

__global__ void scale_by_clip_kernel(float* g, int64_t n, const float* sumsq, float clip)
{
    const float s = clip / fmaxf(sqrtf(*sumsq), clip);          // tf.clip_by_global_norm
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) g[i] = g[i] * s;
}

__global__ void tanh_bwd_kernel(const float* y, const float* dy, float* dx, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dx[i] = dy[i] * (1.0f - y[i] * y[i]);
}

typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);

}  // namespace

extern "C" {

int s2vt_create(const s2vt_dims* d, int32_t max_B, int32_t max_K, s2vt_handle** out)
{
    if (!dims_ok(d) || max_B <= 0 || max_K < 0 || !out) return S2VT_E_BADARG;
    s2vt_handle* h = new (std::nothrow) s2vt_handle;
    if (!h) return S2VT_E_BADARG;
    h->dims = *d; h->max_B = max_B; h->max_K = max_K; h->enc_B = 0; h->ws = nullptr;
    h->ws_bytes = s2vt_sample_workspace_bytes(d, max_B, max_K, 1);
    hipError_t e = hipMalloc(&h->ws, h->ws_bytes);
    if (e != hipSuccess) { delete h; return hip_fail(e); }
    *out = h;
    return S2VT_OK;
}

int s2vt_destroy(s2vt_handle* h)
{
    if (!h) return S2VT_OK;
    hipError_t e = h->ws ? hipFree(h->ws) : hipSuccess;
    delete h;
    return e == hipSuccess ? S2VT_OK : hip_fail(e);
}

int s2vt_encode_fwd(s2vt_handle* h, const s2vt_params* p, const float* video, int32_t B, s2vt_stream stream)
{
    if (!h || !sampler_params_ok(p) || !video || B <= 0 || B > h->max_B) return S2VT_E_BADARG;
    if (chain_fault()) return S2VT_E_CHAIN_TIMEOUT;
    Carver c(h->ws, h->ws_bytes);
    SampleWs w;
    carve_sample(c, &h->dims, B, (h->max_K + 1) * B, &w);
    if (!c.ok()) return S2VT_E_WORKSPACE;
    int rc = sample_encode(&h->dims, p, video, B, w, stream);
    h->enc_B = rc == S2VT_OK ? B : 0;
    return rc;
}

static int decode_common(s2vt_handle* h, const s2vt_params* p, int K, int greedy, uint64_t seed, int video_base, int32_t* ids_out,
                         s2vt_stream stream)
{
    if (!h || !sampler_params_ok(p) || !ids_out || h->enc_B <= 0 || K < 0 || K > h->max_K) return S2VT_E_BADARG;
    if (chain_fault()) return S2VT_E_CHAIN_TIMEOUT;          // the encode this decode continues may be the one that was starved
    Carver c(h->ws, h->ws_bytes);
    SampleWs w;
    carve_sample(c, &h->dims, h->enc_B, (h->max_K + 1) * h->enc_B, &w);     // same layout as the encode call
    return sample_decode(&h->dims, p, h->enc_B, K, greedy, seed, video_base, ids_out, w, stream);
}

int s2vt_decode_greedy(s2vt_handle* h, const s2vt_params* p, int32_t* ids_out, s2vt_stream stream)
{
    return decode_common(h, p, 0, 1, 0, 0, ids_out, stream);
}

int s2vt_decode_multinomial(s2vt_handle* h, const s2vt_params* p, int32_t K, uint64_t seed, int32_t video_base,
                            int32_t* ids_out, s2vt_stream stream)
{
    if (K <= 0) return S2VT_E_BADARG;
    return decode_common(h, p, K, 0, seed, video_base, ids_out, stream);
}

int s2vt_pack_weights(const float* W_tf, int32_t in_dim, int32_t H, float* Wx_packed, float* Wh_packed, s2vt_stream stream)
{
    if (!W_tf || !Wh_packed || in_dim < 0 || H <= 0 || (in_dim > 0 && !Wx_packed)) return S2VT_E_BADARG;
    const size_t nx = (size_t)in_dim * 4 * H, nh = (size_t)H * 4 * H;
    if (nx) hipLaunchKernelGGL(pack_lstm_kernel, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, S(stream), W_tf, in_dim, H, Wx_packed, 1);
    hipLaunchKernelGGL(pack_lstm_kernel, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, S(stream), W_tf + nx, H, H, Wh_packed, 1);
    HIP_TRY(hipGetLastError());
    return S2VT_OK;
}

int s2vt_unpack_weights(const float* Wx_packed, const float* Wh_packed, int32_t in_dim, int32_t H, float* W_tf, s2vt_stream stream)
{
    if (!W_tf || !Wh_packed || in_dim < 0 || H <= 0 || (in_dim > 0 && !Wx_packed)) return S2VT_E_BADARG;
    const size_t nx = (size_t)in_dim * 4 * H, nh = (size_t)H * 4 * H;
    if (nx) hipLaunchKernelGGL(pack_lstm_kernel, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, S(stream), Wx_packed, in_dim, H, W_tf, 0);
    hipLaunchKernelGGL(pack_lstm_kernel, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, S(stream), Wh_packed, H, H, W_tf + nx, 0);
    HIP_TRY(hipGetLastError());
    return S2VT_OK;
}

int s2vt_frame_embed_bwd(const s2vt_dims* d, const float* video, const float* d_emb, int32_t B, float* d_encode_image_W,
                         float* d_encode_image_b, s2vt_stream stream)
{
    if (!dims_ok(d) || !video || !d_emb || !d_encode_image_W || !d_encode_image_b || B < 0) return S2VT_E_BADARG;
    if (B == 0) return S2VT_OK;
    const int R = B * d->n_video_lstm_step, D = d->dim_image, E = d->word_dim;
    TnArgs a{video, nullptr, D, d_emb, E, d_encode_image_W, E, R, D, E, 1};
    HIP_TRY(launch_gemm_tn(a, S(stream)));
    HIP_TRY(launch_colsum(d_emb, E, R, E, d_encode_image_b, S(stream)));
    return S2VT_OK;
}

int s2vt_lstm_cell_bwd(const float* gates, const float* c_new, const float* c_prev, const float* dh, const float* dc_in,
                       float* dz, float* dc_prev, int32_t M, int32_t H, s2vt_stream stream)
{
    if (!gates || !c_new || !dh || !dz || !dc_prev || M < 0 || H <= 0) return S2VT_E_BADARG;
    if (M == 0) return S2VT_OK;
    HIP_TRY(launch_lstm_bwd_pointwise(gates, c_new, c_prev, nullptr, 0, 0, dh, H, dc_in, dc_prev, dz, M, H, 1.0f, 0, 0u, nullptr,
                                      nullptr, S(stream)));
    return S2VT_OK;
}

int s2vt_xent_smooth_fwd_bwd(float* logits, int32_t ld, int32_t R, int32_t V, const int32_t* target, const float* coef,
                             float label_smoothing, float* nll, s2vt_stream stream)
{
    return s2vt_softmax_nll_fwd_bwd(logits, ld, R, V, target, coef, label_smoothing, nll, nullptr, stream);
}

int s2vt_pg_nll_fwd_bwd(float* logits, int32_t ld, int32_t N, int32_t Tc, int32_t V, const int32_t* target_tm,
                        const float* adv, const float* mask, float* coef_scratch, float* nll, float* lp_target,
                        s2vt_stream stream)
{
    if (!logits || !target_tm || !adv || !mask || !coef_scratch || N < 0 || Tc <= 0) return S2VT_E_BADARG;
    if (N == 0) return S2VT_OK;
    hipLaunchKernelGGL(pg_coef_kernel, dim3((N * Tc + 255) / 256), dim3(256), 0, S(stream), adv, mask, coef_scratch, N, Tc);
    HIP_TRY(hipGetLastError());
    return s2vt_softmax_nll_fwd_bwd(logits, ld, N * Tc, V, target_tm, coef_scratch, 0.0f, nll, lp_target, stream);
}

int s2vt_embed_gather(const float* Wemb, int32_t ldw, const int32_t* idx, int32_t R, int32_t E, float* out, int32_t ldo,
                      s2vt_stream stream)
{
    if (!Wemb || !idx || !out || R < 0 || E <= 0 || ldw < E || ldo < E) return S2VT_E_BADARG;
    if (R == 0) return S2VT_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, S(stream), Wemb, ldw, idx, R, E, out, ldo);
    HIP_TRY(hipGetLastError());
    return S2VT_OK;
}

int s2vt_global_norm_clip(float* g, int64_t n, float clip_norm, float* sumsq_scratch, s2vt_stream stream)
{
    if (!g || !sumsq_scratch || n < 0 || !(clip_norm > 0.0f)) return S2VT_E_BADARG;
    HIP_TRY(hipMemsetAsync(sumsq_scratch, 0, sizeof(float), S(stream)));
    HIP_TRY(launch_grad_finalize(g, nullptr, n, nullptr, 0.0f, sumsq_scratch, S(stream)));
    if (n > 0) {
        int blocks = (int)((n + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(scale_by_clip_kernel, dim3(blocks), dim3(256), 0, S(stream), g, n, sumsq_scratch, clip_norm);
        HIP_TRY(hipGetLastError());
    }
    return S2VT_OK;
}

/* ---- generic pieces for graphs composed on the host (the attention captioner's backward) ---- */
int s2vt_gemm_tn(const float* A, int32_t lda, const int32_t* rowidx, const float* Bm, int32_t ldb, float* Cm, int32_t ldc,
                 int32_t Mred, int32_t Kout, int32_t N, int32_t accumulate, s2vt_stream stream)
{
    if (!A || !Bm || !Cm || Mred < 0 || Kout <= 0 || N <= 0 || lda < Kout || ldb < N || ldc < N) return S2VT_E_BADARG;
    if (Mred == 0) return S2VT_OK;
    TnArgs a{A, rowidx, lda, Bm, ldb, Cm, ldc, Mred, Kout, N, accumulate ? 1 : 0};
    HIP_TRY(launch_gemm_tn(a, S(stream)));
    return S2VT_OK;
}

int s2vt_transpose(const float* in, int32_t ldi, float* out, int32_t ldo, int32_t R, int32_t Cc, s2vt_stream stream)
{
    if (!in || !out || R < 0 || Cc < 0 || ldi < Cc || ldo < R) return S2VT_E_BADARG;
    HIP_TRY(launch_transpose(in, ldi, out, ldo, R, Cc, S(stream)));
    return S2VT_OK;
}

int s2vt_colsum(const float* X, int32_t ld, int32_t M, int32_t N, float* out, s2vt_stream stream)
{
    if (!X || !out || M < 0 || N <= 0 || ld < N) return S2VT_E_BADARG;
    HIP_TRY(launch_colsum(X, ld, M, N, out, S(stream)));
    return S2VT_OK;
}

int s2vt_tanh_bwd(const float* y, const float* dy, float* dx, int64_t n, s2vt_stream stream)
{
    if (!y || !dy || !dx || n < 0) return S2VT_E_BADARG;
    if (n == 0) return S2VT_OK;
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), y, dy, dx, n);
    HIP_TRY(hipGetLastError());
    return S2VT_OK;
}

int s2vt_dropout_bwd(const float* dout, int32_t ld, float* dh, int32_t M, int32_t H, float keep, uint64_t seed,
                     uint32_t drop_code, const int32_t* video_id, const int32_t* sample_id, s2vt_stream stream)
{
    if (!dout || !dh || M < 0 || H <= 0 || ld < H || !(keep > 0.0f) || (keep < 1.0f && (!video_id || !sample_id))) return S2VT_E_BADARG;
    HIP_TRY(launch_reduce_dropout(dout, ld, dh, 1, M, M, H, keep, seed, drop_code, video_id, sample_id, S(stream)));
    return S2VT_OK;
}

size_t s2vt_lstm_recurrence_scratch_bytes(int32_t H)
{
    if (H <= 0) return 0;
    return ((chain_scratch_floats(H) * 4 + 255) & ~size_t(255)) + ((kChainSyncBytes + 255) & ~size_t(255));
}

int s2vt_lstm_recurrence_fwd(const float* W, int32_t kw0, const float* b, const float* cinit, int64_t cinit_tstride, int32_t ldcinit,
                             int32_t cinit_steps, float* C_hist, float* H_hist, float* gates, float* out, int32_t M, int32_t H,
                             int32_t T, float keep, uint64_t seed, const int32_t* video_id, const int32_t* sample_id,
                             uint32_t drop_code0, int32_t persistent, void* scratch, size_t scratch_bytes, s2vt_stream stream)
{
    if (!W || !b || !C_hist || !H_hist || M <= 0 || H <= 0 || T < 0 || kw0 < 0 || persistent < -1 || persistent > 1) return S2VT_E_BADARG;
    if (cinit && (cinit_steps < 0 || ldcinit < 4 * H)) return S2VT_E_BADARG;
    if (!(keep > 0.0f) || (keep < 1.0f && (!out || !video_id || !sample_id))) return S2VT_E_BADARG;
    float* abuf = nullptr;
    unsigned* sync = nullptr;
    if (chain_fault()) return S2VT_E_CHAIN_TIMEOUT;
    if (persistent != 0) {
        if (persistent == 1 && !chain_eligible(M, H)) return S2VT_E_BADARG;
        if (persistent == 1 && !chain_operands_ok(W, 4 * H, W)) return S2VT_E_ALIGN;       // (-1 falls back to per-step launches)
        if (chain_eligible(M, H)) {
            if (!scratch) return S2VT_E_BADARG;
            if (reinterpret_cast<uintptr_t>(scratch) & 255u) return S2VT_E_ALIGN;
            Carver c(scratch, scratch_bytes);
            sync = c.take<unsigned>(kChainSyncBytes / 4);
            abuf = c.take<float>(chain_scratch_floats(H));
            if (!c.ok()) return S2VT_E_WORKSPACE;
        }
    }
    NoiseIds ids{video_id, sample_id, seed};
    const size_t MH = (size_t)M * H;
    HIP_TRY(lstm_recurrence(W, kw0, b, cinit, (size_t)cinit_tstride, ldcinit, cinit ? cinit_steps : 0, C_hist, H_hist, MH, gates, 4 * MH, out,
                            MH, M, H, T, keep, ids, drop_code0, abuf, sync, S(stream)));
    return S2VT_OK;
}

int s2vt_chain_timeouts(void) { return (int)chain_timeouts(); }

int s2vt_chain_fault(void) { return chain_fault() ? 1 : 0; }

int s2vt_chain_hold(int on)
{
    chain_hold(on != 0);
    return S2VT_OK;
}

int s2vt_chain_ack(int disable_persistent)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(chain_ack(disable_persistent != 0));
    return S2VT_OK;
}

// ncclAllReduce of the RCCL instance that OWNS the caller's communicator.  A communicator must never be handed to a
// second copy of the library, so nothing is ever loaded here: (1) an entry point registered by the host
// (s2vt_set_rccl_allreduce -- needed when the host loaded its RCCL with RTLD_LOCAL, as torch does), else (2) a symbol
// already visible in the global scope, else (3) an ALREADY LOADED librccl (RTLD_NOLOAD), else an error.
static std::atomic<void*> g_rccl_allreduce{nullptr};

int s2vt_set_rccl_allreduce(void* nccl_allreduce_fn_ptr)
{
    g_rccl_allreduce.store(nccl_allreduce_fn_ptr);
    return S2VT_OK;
}

int s2vt_allreduce_grads(float* bucket, int64_t n, void* rccl_comm, s2vt_stream stream)
{
    if (!bucket || n < 0 || !rccl_comm) return S2VT_E_BADARG;
    void* sym = g_rccl_allreduce.load();
    if (!sym) sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");
    if (!sym) {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            void* lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (lib && (sym = dlsym(lib, "ncclAllReduce"))) break;
        }
    }
    if (!sym) return S2VT_E_BADARG;              // no RCCL in this process: the communicator cannot be genuine
    if (n == 0) return S2VT_OK;
    // ncclDataType_t ncclFloat32 = 7, ncclRedOp_t ncclSum = 0 (rccl.h; part of the NCCL 2.x ABI)
    const int rc = reinterpret_cast<nccl_allreduce_fn>(sym)(bucket, bucket, (size_t)n, 7, 0, rccl_comm, S(stream));
    return rc == 0 ? S2VT_OK : S2VT_E_HIP;
}

}  // extern "C"
