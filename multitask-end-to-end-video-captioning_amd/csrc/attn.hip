// attn.hip -- temporal attention score / softmax / context (original_attention.py:106-128) and the
// multitask attribute head (reinforce_multitask_e2e_attribute_loss.py:375-380), forward + backward.
//
// Forward numerics follow the contract: the score e[t,b] = sum_h tanh(hWa[b,h] + P[t,b,h]) * w[h] is an
// ascending-h fmaf chain, evaluated on the matrix pipe (v_mfma_f32_16x16x4_f32 with the tanh computed
// on the fly as the A fragment and w in column 0 of the B fragment) so it is bit-identical to the
// oracle's sequential chain; exp / divide are the fixed sequences of detmath.h; the context is an
// ascending-t fmaf chain.  Backward is order-free fp32.
#include <hip/hip_runtime.h>

#include "api_util.h"
#include "detmath.h"

using namespace s2vt_api;

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

// one wave per 16 rows of the flattened [Tv*B] score vector (row = t*B + b)
__global__ __launch_bounds__(64) void attn_score_kernel(const float* hWa, const float* P, const float* w, float* e, int TvB,
                                                        int B, int H)
{
    const int lane = threadIdx.x, l15 = lane & 15, lq = lane >> 4;
    const int row = blockIdx.x * 16 + l15;
    const bool ok = row < TvB;
    const int b = ok ? row % B : 0;
    const float* hp = hWa + (size_t)b * H;
    const float* pp = P + (size_t)(ok ? row : 0) * H;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < H; k0 += 4) {
        const int k = k0 + lq;
        float a = 0.f, bw = 0.f;
        if (k < H) {
            if (ok) a = dm_tanhf(hp[k] + pp[k]);
            if (l15 == 0) bw = w[k];
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw, acc, 0, 0, 0);
    }
    // D[i][j]: j = lane & 15, i = 4*(lane>>4) + r  -> column 0 lives in lanes 0,16,32,48
    if (l15 == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = blockIdx.x * 16 + lq * 4 + r;
            if (rr < TvB) e[rr] = acc[r];
        }
    }
}

// alpha = exp(e) / (sum_t exp(e) (+1 if 0)) ; ctx[b,h] = chain_t alpha[t,b] * V[t,b,h]
__global__ __launch_bounds__(256) void attn_softmax_ctx_kernel(const float* e, const float* Vt, float* alpha, float* ctx,
                                                               int Tv, int B, int H)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, h = i % H;
    float den = 0.f;
    for (int t = 0; t < Tv; ++t) den = den + dm_expf(e[t * B + b]);
    if (den == 0.f) den = den + 1.0f;
    float c = 0.f;
    for (int t = 0; t < Tv; ++t) {
        const float a = dm_expf(e[t * B + b]) / den;
        if (h == 0) alpha[t * B + b] = a;
        c = __builtin_fmaf(a, Vt[((size_t)t * B + b) * H + h], c);
    }
    ctx[i] = c;
}

// ---- backward -------------------------------------------------------------------------------
// per b: dalpha[t] = sum_h dctx[b,h]*V[t,b,h];  de[t] = alpha[t]*(dalpha[t] - sum_t' alpha[t']*dalpha[t'])
__global__ __launch_bounds__(256) void attn_bwd_de_kernel(const float* dctx, const float* Vt, const float* alpha, float* de,
                                                          int Tv, int B, int H)
{
    __shared__ float sh[4];
    __shared__ float dal[64];
    const int b = blockIdx.x;
    for (int t = 0; t < Tv; ++t) {
        float s = 0.f;
        for (int h = threadIdx.x; h < H; h += 256) s += dctx[(size_t)b * H + h] * Vt[((size_t)t * B + b) * H + h];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) dal[t] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float dot = 0.f;
        for (int t = 0; t < Tv; ++t) dot += alpha[t * B + b] * dal[t];
        for (int t = 0; t < Tv; ++t) de[t * B + b] = alpha[t * B + b] * (dal[t] - dot);
    }
}

// dS = de*w*(1-T^2) -> dP[t,b,h] ; dhWa[b,h] = sum_t dS ; dV[t,b,h] (+)= alpha*dctx ; dw[h] += sum de*T
__global__ __launch_bounds__(256) void attn_bwd_main_kernel(const float* hWa, const float* P, const float* w, const float* alpha,
                                                            const float* de, const float* dctx, float* dhWa, float* dP,
                                                            float* dVt, float* dw, int Tv, int B, int H)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, h = i % H;
    float acc = 0.f, dwl = 0.f;
    const float dc = dctx[i], wh = w[h], hv = hWa[i];
    for (int t = 0; t < Tv; ++t) {
        const size_t o = ((size_t)t * B + b) * H + h;
        const float T = dm_tanhf(hv + P[o]);
        const float d = de[t * B + b];
        const float ds = d * wh * (1.f - T * T);
        dP[o] = ds;
        acc += ds;
        dVt[o] = alpha[t * B + b] * dc;
        dwl += d * T;
    }
    dhWa[i] = acc;
    atomicAdd(dw + h, dwl);
}

// ---- attribute head ---------------------------------------------------------------------------
__global__ void mean_frames_kernel(const float* video, float* out, int B, int Tv, int D)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    const int b = i / D, d = i % D;
    float s = 0.f;
    for (int t = 0; t < Tv; ++t) s = s + video[((size_t)b * Tv + t) * D + d];
    out[i] = s / (float)Tv;
}

// bce = max(z,0) - z*y + log(1 + exp(-|z|)) ; dz = scale * (sigmoid(z) - y)
__global__ void sigmoid_bce_kernel(const float* z, const float* y, float* bce, float* dz, float scale, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float zz = z[i], yy = y[i];
    const float az = zz < 0.f ? -zz : zz;
    const float sp = dm_logf(1.0f + dm_expf(-az));
    if (bce) bce[i] = ((zz > 0.f ? zz : 0.f) - zz * yy) + sp;
    if (dz) dz[i] = scale * (dm_sigmoidf(zz) - yy);
}

}  // namespace

extern "C" {

int s2vt_attention_fwd(const float* hWa, const float* P, const float* Vt, const float* w, float* scores, float* alpha,
                       float* ctx, int32_t Tv, int32_t B, int32_t H, s2vt_stream stream)
{
    if (!hWa || !P || !Vt || !w || !scores || !alpha || !ctx || Tv <= 0 || Tv > 64 || B <= 0 || H <= 0) return S2VT_E_BADARG;
    hipStream_t st = S(stream);
    hipLaunchKernelGGL(attn_score_kernel, dim3((Tv * B + 15) / 16), dim3(64), 0, st, hWa, P, w, scores, Tv * B, B, H);
    hipLaunchKernelGGL(attn_softmax_ctx_kernel, dim3((B * H + 255) / 256), dim3(256), 0, st, scores, Vt, alpha, ctx, Tv, B, H);
    HIP_TRY(hipGetLastError());
    return S2VT_OK;
}

int s2vt_attention_bwd(const float* hWa, const float* P, const float* Vt, const float* w, const float* alpha,
                       const float* dctx, float* de_scratch, float* dhWa, float* dP, float* dVt, float* dw, int32_t Tv,
                       int32_t B, int32_t H, s2vt_stream stream)
{
    if (!hWa || !P || !Vt || !w || !alpha || !dctx || !de_scratch || !dhWa || !dP || !dVt || !dw || Tv <= 0 || Tv > 64 ||
        B <= 0 || H <= 0)
        return S2VT_E_BADARG;
    hipStream_t st = S(stream);
    hipLaunchKernelGGL(attn_bwd_de_kernel, dim3(B), dim3(256), 0, st, dctx, Vt, alpha, de_scratch, Tv, B, H);
    hipLaunchKernelGGL(attn_bwd_main_kernel, dim3((B * H + 255) / 256), dim3(256), 0, st, hWa, P, w, alpha, de_scratch, dctx,
                       dhWa, dP, dVt, dw, Tv, B, H);
    HIP_TRY(hipGetLastError());
    return S2VT_OK;
}

int s2vt_attr_head_fwd(const float* video, int32_t B, int32_t Tv, int32_t D, const float* attr_W, const float* attr_b,
                       int32_t A, const float* labels, float* mean_feat, float* z, float* bce, s2vt_stream stream)
{
    if (!video || !attr_W || !attr_b || !mean_feat || !z || B <= 0 || Tv <= 0 || D <= 0 || A <= 0) return S2VT_E_BADARG;
    hipStream_t st = S(stream);
    hipLaunchKernelGGL(mean_frames_kernel, dim3((B * D + 255) / 256), dim3(256), 0, st, video, mean_feat, B, Tv, D);
    HIP_TRY(hipGetLastError());
    ASeg a = make_seg(mean_feat, D, D, 0);
    HIP_TRY(store_call(&a, 1, attr_W, A, attr_b, z, A, B, A, 0, -1, st));
    if (labels && bce) {
        hipLaunchKernelGGL(sigmoid_bce_kernel, dim3((B * A + 255) / 256), dim3(256), 0, st, z, labels, bce, nullptr, 0.f, B * A);
        HIP_TRY(hipGetLastError());
    }
    return S2VT_OK;
}

int s2vt_attr_head_bwd(const float* mean_feat, const float* z, const float* labels, int32_t B, int32_t D, int32_t A,
                       float scale, float* dz_scratch, float* d_attr_W, float* d_attr_b, s2vt_stream stream)
{
    if (!mean_feat || !z || !labels || !dz_scratch || !d_attr_W || !d_attr_b || B <= 0 || D <= 0 || A <= 0) return S2VT_E_BADARG;
    hipStream_t st = S(stream);
    hipLaunchKernelGGL(sigmoid_bce_kernel, dim3((B * A + 255) / 256), dim3(256), 0, st, z, labels, nullptr, dz_scratch, scale,
                       B * A);
    HIP_TRY(hipGetLastError());
    TnArgs t{mean_feat, nullptr, D, dz_scratch, A, d_attr_W, A, B, D, A, 1};
    HIP_TRY(launch_gemm_tn(t, st));
    HIP_TRY(launch_colsum(dz_scratch, A, B, A, d_attr_b, st));
    return S2VT_OK;
}

}  // extern "C"
