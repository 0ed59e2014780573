// attn.hip -- temporal attention score / softmax / context (original_attention.py:106-128) and the
// multitask attribute head (reinforce_multitask_e2e_attribute_loss.py:375-380), forward + backward.
//
// Forward numerics follow the contract: the score e[t,b] = sum_h tanh(hWa[b,h] + P[t,b,h]) * w[h] is an
// ascending-h fmaf chain (one VALU lane per frame runs it: v_fma_f32 is the oracle's fmaf), exp / divide are the
// fixed sequences of detmath.h, the context is an ascending-t fmaf chain: bit-identical to the oracle's
// orc_attention_step.  One launch per decode step does score -> softmax -> context (one workgroup per batch row).
// Backward is order-free fp32, one launch per step as well.  The whole-model entry points are in attn_model.hip.
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>

#include "api_util.h"
#include "detmath.h"

using namespace s2vt_api;

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

// ---- forward: one workgroup per batch row b does the whole step for that row -------------------------------------
//   phase 1  T[t][h] = tanh(hWa[b,h] + P[t,b,h]) for up to RC frames at a time, all 256 threads, into LDS
//   phase 2  e[t] = the ascending-h fmaf chain sum_h T[t][h] * w[h]: ONE lane per frame runs the 1000-long chain on the
//            VALU (v_fma_f32 is the same fused multiply-add the oracle's fmaf is; a dependent VALU chain issues every
//            ~4-5 cycles per link where the 16x16x4 MFMA form of round 3 took 40 cycles per four links and used 1/16 of
//            each MFMA), frames spread over the four waves so that the four SIMDs run them side by side
//   phase 3  alpha = exp(e) / (sum_t exp(e) (+1 if 0)) in ascending t, the first-8-frames sum of the regulariser
//   phase 4  ctx[b,h] = ascending-t fmaf chain of alpha[t] * V[t,b,h]
// Bit-identical to oracle/s2vt_oracle.c::orc_attention_step.
constexpr int kAttnMaxTv = 64;
constexpr int kAttnChunkRows = 32;       // frames whose tanh rows are resident in LDS at a time (32 x 1004 floats = 126 KB at H = 1000)

__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnFwdArgs a)
{
    extern __shared__ float sm[];
    const int H = a.H, Tv = a.Tv, B = a.B, ldT = a.ldT;
    float* Tt = sm;                                  // [RC][ldT]
    float* wl = sm + (size_t)a.RC * ldT;             // [ldT] the score vector w
    float* ev = wl + ldT;                            // [64] scores
    float* xv = ev + kAttnMaxTv;                     // [64] exp(e), then alpha
    float* sc = xv + kAttnMaxTv;                     // [4] scalars
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* hp = a.hWa ? a.hWa + (size_t)b * H : nullptr;
    for (int h = tid; h < H; h += 256) wl[h] = a.w[h];
    for (int c0 = 0; c0 < Tv; c0 += a.RC) {
        const int nr = (Tv - c0) < a.RC ? (Tv - c0) : a.RC;
        if (a.vec) {
            const int H4 = H >> 2;
            for (int i = tid; i < nr * H4; i += 256) {
                const int r = i / H4, q = i - r * H4;
                float4 hv = hp ? *reinterpret_cast<const float4*>(hp + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 pv = *reinterpret_cast<const float4*>(a.P + ((size_t)(c0 + r) * B + b) * H + 4 * q);
                float4 t;
                t.x = dm_tanhf(hv.x + pv.x); t.y = dm_tanhf(hv.y + pv.y); t.z = dm_tanhf(hv.z + pv.z); t.w = dm_tanhf(hv.w + pv.w);
                *reinterpret_cast<float4*>(Tt + (size_t)r * ldT + 4 * q) = t;
            }
        } else {
            for (int i = tid; i < nr * H; i += 256) {
                const int r = i / H, h = i - r * H;
                Tt[(size_t)r * ldT + h] = dm_tanhf((hp ? hp[h] : 0.f) + a.P[((size_t)(c0 + r) * B + b) * H + h]);
            }
        }
        __syncthreads();
        const int r = (tid & 63) * 4 + (tid >> 6);       // frame r of the chunk -> wave r & 3, lane r >> 2
        if (r < nr) {
            const float* tr = Tt + (size_t)r * ldT;
            float e = 0.f;
            int h = 0;
#pragma unroll 4
            for (; h + 4 <= H; h += 4) {
                const float4 t = *reinterpret_cast<const float4*>(tr + h);
                const float4 ww = *reinterpret_cast<const float4*>(wl + h);
                e = __builtin_fmaf(t.x, ww.x, e);
                e = __builtin_fmaf(t.y, ww.y, e);
                e = __builtin_fmaf(t.z, ww.z, e);
                e = __builtin_fmaf(t.w, ww.w, e);
            }
            for (; h < H; ++h) e = __builtin_fmaf(tr[h], wl[h], e);
            ev[c0 + r] = e;
        }
        __syncthreads();
    }
    if (tid < Tv) {
        const float e = ev[tid];
        if (a.scores) a.scores[tid * B + b] = e;
        xv[tid] = dm_expf(e);
    }
    __syncthreads();
    if (tid == 0) {
        float den = 0.f;
        for (int t = 0; t < Tv; ++t) den = den + xv[t];
        if (den == 0.f) den = den + 1.0f;
        sc[0] = den;
    }
    __syncthreads();
    if (tid < Tv) {
        const float al = xv[tid] / sc[0];
        a.alpha[tid * B + b] = al;
        ev[tid] = al;
    }
    __syncthreads();
    if (tid == 0 && a.asum) {
        float s = 0.f;
        const int n8 = Tv < 8 ? Tv : 8;
        for (int t = 0; t < n8; ++t) s = s + ev[t];
        a.asum[b] = s;
    }
    if (a.vec) {
        for (int q = tid; q < (H >> 2); q += 256) {
            float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int t = 0; t < Tv; ++t) {
                const float al = ev[t];
                const float4 v = *reinterpret_cast<const float4*>(a.Vt + ((size_t)t * B + b) * H + 4 * q);
                c.x = __builtin_fmaf(al, v.x, c.x); c.y = __builtin_fmaf(al, v.y, c.y);
                c.z = __builtin_fmaf(al, v.z, c.z); c.w = __builtin_fmaf(al, v.w, c.w);
            }
            *reinterpret_cast<float4*>(a.ctx + (size_t)b * H + 4 * q) = c;
        }
    } else {
        for (int h = tid; h < H; h += 256) {
            float c = 0.f;
            for (int t = 0; t < Tv; ++t) c = __builtin_fmaf(ev[t], a.Vt[((size_t)t * B + b) * H + h], c);
            a.ctx[(size_t)b * H + h] = c;
        }
    }
}

// ---- backward (order-free fp32), one workgroup per batch row ----------------------------------------------------------
//   dctx[h]   = dense part + sum of the split-K slabs the caller's product left (optional)
//   dalpha[t] = sum_h dctx[h] * V[t,b,h]   (one wave per frame, shuffle reduction)
//   regulariser beta * max(0, m - sum(alpha[0:8])) * mask (original_attention.py:123,144): dalpha[t < 8] -= reg_coef[b] while
//               the hinge is open (m - asum[b] > 0)
//   de[t]     = alpha[t] * (dalpha[t] - sum_t' alpha[t'] dalpha[t'])
//   dS = de * w * (1 - T^2) -> dP[t,b,h] ; dhWa[b,h] = sum_t dS ; dV[t,b,h] = alpha * dctx ; dw[h] += sum_t de * T
// dP / dV are accumulated in place over the decode steps when acc != 0 (the workgroup owns its rows).
__global__ __launch_bounds__(256) void attn_bwd_kernel(const AttnBwdArgs a)
{
    extern __shared__ float sm[];
    const int H = a.H, Tv = a.Tv, B = a.B;
    float* dc = sm;                      // [H]
    float* dal = sm + ((H + 3) & ~3);    // [64]
    float* de = dal + kAttnMaxTv;        // [64]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int h = tid; h < H; h += 256) {
        float v = a.dctx ? a.dctx[(size_t)b * a.ld_dctx + h] : 0.f;
        for (int s = 0; s < a.nslab; ++s) v += a.slabs[(size_t)s * a.slab_stride + (size_t)b * a.ld_slab + a.ctx_col0 + h];
        dc[h] = v;
        if (a.demb_out) {
            float u = a.demb_dense ? a.demb_dense[(size_t)b * a.ld_demb + h] : 0.f;
            for (int s = 0; s < a.nslab; ++s) u += a.slabs[(size_t)s * a.slab_stride + (size_t)b * a.ld_slab + a.emb_col0 + h];
            a.demb_out[(size_t)b * H + h] = u;
        }
    }
    __syncthreads();
    for (int t = wv; t < Tv; t += 4) {
        const float* vp = a.Vt + ((size_t)t * B + b) * H;
        float s = 0.f;
        for (int h = lane; h < H; h += 64) s += dc[h] * vp[h];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) dal[t] = s;
    }
    __syncthreads();
    if (tid == 0) {
        if (a.reg_coef && (a.reg_m - a.asum[b]) > 0.f) {
            const int n8 = Tv < 8 ? Tv : 8;
            const float rc = a.reg_coef[b];
            for (int t = 0; t < n8; ++t) dal[t] -= rc;
        }
        float dot = 0.f;
        for (int t = 0; t < Tv; ++t) dot += a.alpha[t * B + b] * dal[t];
        for (int t = 0; t < Tv; ++t) de[t] = a.alpha[t * B + b] * (dal[t] - dot);
    }
    __syncthreads();
    for (int h = tid; h < H; h += 256) {
        const float hv = a.hWa ? a.hWa[(size_t)b * H + h] : 0.f, wh = a.w[h], dch = dc[h];
        float acc = 0.f, dwl = 0.f;
        for (int t = 0; t < Tv; ++t) {
            const size_t o = ((size_t)t * B + b) * H + h;
            const float T = dm_tanhf(hv + a.P[o]);
            const float d = de[t];
            const float ds = d * wh * (1.f - T * T);
            const float dv = a.alpha[t * B + b] * dch;
            if (a.acc) { a.dP[o] += ds; a.dVt[o] += dv; }
            else { a.dP[o] = ds; a.dVt[o] = dv; }
            acc += ds;
            dwl += d * T;
        }
        if (a.dhWa) a.dhWa[(size_t)b * H + h] = acc;
        atomicAdd(a.dw + h, dwl);
    }
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

namespace s2vt {

hipError_t launch_attn_fwd(const AttnFwdArgs& a0, hipStream_t st)
{
    if (a0.Tv <= 0 || a0.Tv > kAttnMaxTv || a0.B <= 0 || a0.H <= 0) return hipErrorInvalidValue;
    AttnFwdArgs a = a0;
    a.RC = a.Tv < kAttnChunkRows ? a.Tv : kAttnChunkRows;
    a.ldT = ((a.H + 3) & ~3) + 4;
    const auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    a.vec = (!(a.H & 3) && al16(a.P) && al16(a.Vt) && al16(a.ctx) && (!a.hWa || al16(a.hWa))) ? 1 : 0;
    const size_t lds = ((size_t)(a.RC + 1) * a.ldT + 2 * kAttnMaxTv + 4) * sizeof(float);
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (attr_err != hipSuccess) return attr_err;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(a.B), dim3(256), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_attn_bwd(const AttnBwdArgs& a, hipStream_t st)
{
    if (a.Tv <= 0 || a.Tv > kAttnMaxTv || a.B <= 0 || a.H <= 0) return hipErrorInvalidValue;
    const size_t lds = ((size_t)((a.H + 3) & ~3) + 2 * kAttnMaxTv) * sizeof(float);
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(a.B), dim3(256), lds, st, a);
    return hipGetLastError();
}

}  // namespace s2vt

extern "C" {

int s2vt_attention_fwd(const float* hWa, const float* P, const float* Vt, const float* w, float* scores, float* alpha,
                       float* ctx, int32_t Tv, int32_t B, int32_t H, s2vt_stream stream)
{
    if (!hWa || !P || !Vt || !w || !scores || !alpha || !ctx || Tv <= 0 || Tv > kAttnMaxTv || B <= 0 || H <= 0) return S2VT_E_BADARG;
    AttnFwdArgs a;
    std::memset(&a, 0, sizeof(a));
    a.hWa = hWa; a.P = P; a.Vt = Vt; a.w = w; a.scores = scores; a.alpha = alpha; a.ctx = ctx; a.Tv = Tv; a.B = B; a.H = H;
    HIP_TRY(launch_attn_fwd(a, S(stream)));
    return S2VT_OK;
}

int s2vt_attention_bwd(const float* hWa, const float* P, const float* Vt, const float* w, const float* alpha,
                       const float* dctx, float* de_scratch, float* dhWa, float* dP, float* dVt, float* dw, int32_t Tv,
                       int32_t B, int32_t H, s2vt_stream stream)
{
    if (!hWa || !P || !Vt || !w || !alpha || !dctx || !de_scratch || !dhWa || !dP || !dVt || !dw || Tv <= 0 || Tv > kAttnMaxTv ||
        B <= 0 || H <= 0)
        return S2VT_E_BADARG;
    AttnBwdArgs a;
    std::memset(&a, 0, sizeof(a));
    a.hWa = hWa; a.P = P; a.Vt = Vt; a.w = w; a.alpha = alpha; a.dctx = dctx; a.ld_dctx = H;
    a.dhWa = dhWa; a.dP = dP; a.dVt = dVt; a.dw = dw; a.Tv = Tv; a.B = B; a.H = H;
    HIP_TRY(launch_attn_bwd(a, S(stream)));
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
