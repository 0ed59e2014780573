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
#include "attn_score.h"
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
    const float* __restrict__ hp = a.hWa ? a.hWa + (size_t)b * H : nullptr;
    const float* __restrict__ Pp = a.P;
    const float* __restrict__ Vp = a.Vt;
    const int Hp = (H + 15) & ~15;
    for (int h = tid; h < Hp; h += 256) wl[h] = h < H ? a.w[h] : 0.f;
    for (int i = tid; i < a.RC * (Hp - H); i += 256) Tt[(size_t)(i / (Hp - H)) * ldT + H + i % (Hp - H)] = 0.f;     // pad columns (never written again)
    for (int c0 = 0; c0 < Tv; c0 += a.RC) {
        const int nr = (Tv - c0) < a.RC ? (Tv - c0) : a.RC;
        if (a.vec) {
            // thread = one 16-byte column group, all frames of the chunk: the loads of up to 8 frames are issued together
            // (global latency is paid once per 8 frames, not once per frame)
            for (int q = tid; q < (H >> 2); q += 256) {
                const float4 hv = hp ? *reinterpret_cast<const float4*>(hp + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
                for (int r0 = 0; r0 < nr; r0 += 8) {
                    float4 pv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (r0 + j < nr) pv[j] = *reinterpret_cast<const float4*>(Pp + ((size_t)(c0 + r0 + j) * B + b) * H + 4 * q);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (r0 + j < nr) {
                            float4 t;
                            t.x = dm_tanhf(hv.x + pv[j].x); t.y = dm_tanhf(hv.y + pv[j].y); t.z = dm_tanhf(hv.z + pv[j].z); t.w = dm_tanhf(hv.w + pv[j].w);
                            *reinterpret_cast<float4*>(Tt + (size_t)(r0 + j) * ldT + 4 * q) = t;
                        }
                }
            }
        } else {
            for (int i = tid; i < nr * H; i += 256) {
                const int r = i / H, h = i - r * H;
                Tt[(size_t)r * ldT + h] = dm_tanhf((hp ? hp[h] : 0.f) + Pp[((size_t)(c0 + r) * B + b) * H + h]);
            }
        }
        __syncthreads();
        const int r = (tid & 63) * 4 + (tid >> 6);       // frame r of the chunk -> wave r & 3, lane r >> 2
        if (r < nr) ev[c0 + r] = score_chain(Tt + (size_t)r * ldT, wl, (H + 15) >> 4);     // (rows zero-padded to a multiple of 16: attn_score.h)
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
            for (int t0 = 0; t0 < Tv; t0 += 8) {
                float4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (t0 + j < Tv) v[j] = *reinterpret_cast<const float4*>(Vp + ((size_t)(t0 + j) * B + b) * H + 4 * q);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (t0 + j < Tv) {
                        const float al = ev[t0 + j];
                        c.x = __builtin_fmaf(al, v[j].x, c.x); c.y = __builtin_fmaf(al, v[j].y, c.y);
                        c.z = __builtin_fmaf(al, v[j].z, c.z); c.w = __builtin_fmaf(al, v[j].w, c.w);
                    }
            }
            *reinterpret_cast<float4*>(a.ctx + (size_t)b * H + 4 * q) = c;
        }
    } else {
        for (int h = tid; h < H; h += 256) {
            float c = 0.f;
            for (int t = 0; t < Tv; ++t) c = __builtin_fmaf(ev[t], Vp[((size_t)t * B + b) * H + h], c);
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
    float* al = de + kAttnMaxTv;         // [64]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* __restrict__ Pp = a.P;
    const float* __restrict__ Vp = a.Vt;
    const float* __restrict__ sl = a.slabs;
    float* __restrict__ dPp = a.dP;
    float* __restrict__ dVp = a.dVt;
    if (tid < Tv) al[tid] = a.alpha[tid * B + b];
    // every load of a thread is issued before the first is used (no store in between: nothing forces the compiler to wait)
    if (a.vec) {
        for (int q = tid; q < (H >> 2); q += 256) {
            float4 v = a.dctx ? *reinterpret_cast<const float4*>(a.dctx + (size_t)b * a.ld_dctx + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 u = (a.demb_out && a.demb_dense) ? *reinterpret_cast<const float4*>(a.demb_dense + (size_t)b * a.ld_demb + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            for (int s = 0; s < a.nslab; ++s) {
                const float* row = sl + (size_t)s * a.slab_stride + (size_t)b * a.ld_slab + 4 * q;
                const float4 x = *reinterpret_cast<const float4*>(row + a.ctx_col0);
                v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
                if (a.demb_out) {
                    const float4 y = *reinterpret_cast<const float4*>(row + a.emb_col0);
                    u.x += y.x; u.y += y.y; u.z += y.z; u.w += y.w;
                }
            }
            *reinterpret_cast<float4*>(dc + 4 * q) = v;
            if (a.demb_out) *reinterpret_cast<float4*>(a.demb_out + (size_t)b * H + 4 * q) = u;
        }
    } else {
        for (int h = tid; h < H; h += 256) {
            float v = a.dctx ? a.dctx[(size_t)b * a.ld_dctx + h] : 0.f;
            for (int s = 0; s < a.nslab; ++s) v += sl[(size_t)s * a.slab_stride + (size_t)b * a.ld_slab + a.ctx_col0 + h];
            dc[h] = v;
            if (a.demb_out) {
                float u = a.demb_dense ? a.demb_dense[(size_t)b * a.ld_demb + h] : 0.f;
                for (int s = 0; s < a.nslab; ++s) u += sl[(size_t)s * a.slab_stride + (size_t)b * a.ld_slab + a.emb_col0 + h];
                a.demb_out[(size_t)b * H + h] = u;
            }
        }
    }
    __syncthreads();
    for (int t = wv; t < Tv; t += 4) {
        const float* vp = Vp + ((size_t)t * B + b) * H;
        float s = 0.f;
        if (a.vec) {
            for (int q = lane; q < (H >> 2); q += 64) {
                const float4 x = *reinterpret_cast<const float4*>(vp + 4 * q), d = *reinterpret_cast<const float4*>(dc + 4 * q);
                s += d.x * x.x + d.y * x.y + d.z * x.z + d.w * x.w;
            }
        } else {
            for (int h = lane; h < H; h += 64) s += dc[h] * vp[h];
        }
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
        for (int t = 0; t < Tv; ++t) dot += al[t] * dal[t];
        for (int t = 0; t < Tv; ++t) de[t] = al[t] * (dal[t] - dot);
    }
    __syncthreads();
    if (a.vec) {
        for (int q = tid; q < (H >> 2); q += 256) {
            const float4 hv = a.hWa ? *reinterpret_cast<const float4*>(a.hWa + (size_t)b * H + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 wh = *reinterpret_cast<const float4*>(a.w + 4 * q), dch = *reinterpret_cast<const float4*>(dc + 4 * q);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), dwl = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int t0 = 0; t0 < Tv; t0 += 4) {
                float4 pv[4], op[4], ov[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (t0 + j < Tv) {
                        const size_t o = ((size_t)(t0 + j) * B + b) * H + 4 * q;
                        pv[j] = *reinterpret_cast<const float4*>(Pp + o);
                        if (a.acc) { op[j] = *reinterpret_cast<const float4*>(dPp + o); ov[j] = *reinterpret_cast<const float4*>(dVp + o); }
                    }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (t0 + j < Tv) {
                        const size_t o = ((size_t)(t0 + j) * B + b) * H + 4 * q;
                        const float d = de[t0 + j], alt = al[t0 + j];
                        const float T0 = dm_tanhf(hv.x + pv[j].x), T1 = dm_tanhf(hv.y + pv[j].y), T2 = dm_tanhf(hv.z + pv[j].z), T3 = dm_tanhf(hv.w + pv[j].w);
                        float4 ds, dv;
                        ds.x = d * wh.x * (1.f - T0 * T0); ds.y = d * wh.y * (1.f - T1 * T1); ds.z = d * wh.z * (1.f - T2 * T2); ds.w = d * wh.w * (1.f - T3 * T3);
                        dv.x = alt * dch.x; dv.y = alt * dch.y; dv.z = alt * dch.z; dv.w = alt * dch.w;
                        acc.x += ds.x; acc.y += ds.y; acc.z += ds.z; acc.w += ds.w;
                        dwl.x += d * T0; dwl.y += d * T1; dwl.z += d * T2; dwl.w += d * T3;
                        if (a.acc) {
                            ds.x += op[j].x; ds.y += op[j].y; ds.z += op[j].z; ds.w += op[j].w;
                            dv.x += ov[j].x; dv.y += ov[j].y; dv.z += ov[j].z; dv.w += ov[j].w;
                        }
                        *reinterpret_cast<float4*>(dPp + o) = ds;
                        *reinterpret_cast<float4*>(dVp + o) = dv;
                    }
            }
            if (a.dhWa) *reinterpret_cast<float4*>(a.dhWa + (size_t)b * H + 4 * q) = acc;
            atomicAdd(a.dw + 4 * q, dwl.x); atomicAdd(a.dw + 4 * q + 1, dwl.y); atomicAdd(a.dw + 4 * q + 2, dwl.z); atomicAdd(a.dw + 4 * q + 3, dwl.w);
        }
    } else {
        for (int h = tid; h < H; h += 256) {
            const float hv = a.hWa ? a.hWa[(size_t)b * H + h] : 0.f, wh = a.w[h], dch = dc[h];
            float acc = 0.f, dwl = 0.f;
            for (int t = 0; t < Tv; ++t) {
                const size_t o = ((size_t)t * B + b) * H + h;
                const float T = dm_tanhf(hv + Pp[o]);
                const float d = de[t];
                const float ds = d * wh * (1.f - T * T);
                const float dv = al[t] * dch;
                if (a.acc) { dPp[o] += ds; dVp[o] += dv; }
                else { dPp[o] = ds; dVp[o] = dv; }
                acc += ds;
                dwl += d * T;
            }
            if (a.dhWa) a.dhWa[(size_t)b * H + h] = acc;
            atomicAdd(a.dw + h, dwl);
        }
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

// scores = sigmoid(z): evaluate_multilabel (reinforce_multitask_e2e_attribute_loss.py:624)
__global__ void sigmoid_kernel(const float* z, float* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = dm_sigmoidf(z[i]);
}

}  // namespace

namespace s2vt {

hipError_t launch_attn_fwd(const AttnFwdArgs& a0, hipStream_t st)
{
    if (a0.Tv <= 0 || a0.Tv > kAttnMaxTv || a0.B <= 0 || a0.H <= 0) return hipErrorInvalidValue;
    AttnFwdArgs a = a0;
    a.RC = a.Tv < kAttnChunkRows ? a.Tv : kAttnChunkRows;
    a.ldT = ((a.H + 15) & ~15) + 4;                          // rows padded to whole 16-link phases of the score chain (+4: bank spread)
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
    // launch profiler class 7: attention forward (flops = the score and context chains, 2 x 2 Tv B H; HBM-bound by P and V)
    if (!prof_wants(7, 0)) {
        hipLaunchKernelGGL(attn_fwd_kernel, dim3(a.B), dim3(256), lds, st, a);
        return hipGetLastError();
    }
    hipEvent_t e0, e1;
    hipError_t pe = prof_events(&e0, &e1);
    if (pe != hipSuccess) return pe;
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(a.B), dim3(256), lds, st, a);
    (void)hipEventRecord(e1, st);
    prof_record(7, 0, "attn_fwd(score+softmax+ctx)", 4.0 * a.Tv * a.B * (double)a.H, e0, e1);
    return hipGetLastError();
}

hipError_t launch_attn_bwd(const AttnBwdArgs& a0, hipStream_t st)
{
    if (a0.Tv <= 0 || a0.Tv > kAttnMaxTv || a0.B <= 0 || a0.H <= 0) return hipErrorInvalidValue;
    AttnBwdArgs a = a0;
    {
        const auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
        bool v = !(a.H & 3) && al16(a.P) && al16(a.Vt) && al16(a.w) && al16(a.dP) && al16(a.dVt) && al16(a.dw) && (!a.hWa || al16(a.hWa)) &&
                 (!a.dhWa || al16(a.dhWa)) && (!a.dctx || (al16(a.dctx) && !(a.ld_dctx & 3)));
        if (a.nslab > 0) v = v && al16(a.slabs) && !(a.slab_stride & 3) && !(a.ld_slab & 3) && !(a.ctx_col0 & 3) && !(a.emb_col0 & 3);
        if (a.demb_out) v = v && al16(a.demb_out) && (!a.demb_dense || (al16(a.demb_dense) && !(a.ld_demb & 3)));
        a.vec = v ? 1 : 0;
    }
    const size_t lds = ((size_t)((a.H + 3) & ~3) + 3 * kAttnMaxTv) * sizeof(float);
    if (!prof_wants(8, 0)) {
        hipLaunchKernelGGL(attn_bwd_kernel, dim3(a.B), dim3(256), lds, st, a);
        return hipGetLastError();
    }
    hipEvent_t e0, e1;
    hipError_t pe = prof_events(&e0, &e1);
    if (pe != hipSuccess) return pe;
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(a.B), dim3(256), lds, st, a);
    (void)hipEventRecord(e1, st);
    prof_record(8, 0, "attn_bwd", 10.0 * a.Tv * a.B * (double)a.H, e0, e1);
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

int s2vt_attr_head_scores(const float* video, int32_t B, int32_t Tv, int32_t D, const float* attr_W, const float* attr_b,
                          int32_t A, float* mean_feat, float* z, float* scores, s2vt_stream stream)
{
    if (!scores) return S2VT_E_BADARG;
    const int rc = s2vt_attr_head_fwd(video, B, Tv, D, attr_W, attr_b, A, nullptr, mean_feat, z, nullptr, stream);
    if (rc != S2VT_OK) return rc;
    hipLaunchKernelGGL(sigmoid_kernel, dim3((B * A + 255) / 256), dim3(256), 0, S(stream), z, scores, B * A);
    HIP_TRY(hipGetLastError());
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
