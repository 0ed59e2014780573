// aux.hip -- the HBM-bound kernels around the contractions: softmax/NLL rows, LSTM pointwise
// backward, column sums, embedding scatter-add, transposes, gradient finalisation, TF-form Adam,
// and the TN weight-gradient GEMM launcher.  All 64-lane-wave code, 16-byte accesses where the
// layout allows, one pass over HBM per kernel.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>

#include "detmath.h"
#include "gemm_tn.h"
#include "internal.h"
#include "chain_common.h"

namespace s2vt {

// ---------------------------------------------------------------------------------------------
// wave / block reductions
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
template <bool IS_MAX>
__device__ __forceinline__ float block_reduce(float v, float* sh)
{
    v = IS_MAX ? wave_max(v) : wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = sh[0];
    for (int i = 1; i < nw; ++i) r = IS_MAX ? fmaxf(r, sh[i]) : r + sh[i];
    return r;
}

// ---------------------------------------------------------------------------------------------
// caption [N,Tc] -> time-major previous-token and target arrays
//   prev[t][n] = t == 0 ? <bos>=1 : caption[n][t-1]   (tf_s2vt.py:128-134)
//   tgt [t][n] = caption[n][t]                         (tf_s2vt.py:150)
// ---------------------------------------------------------------------------------------------
__global__ void prep_caption_kernel(const int32_t* cap, int32_t* prev, int32_t* tgt, int N, int Tc)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * Tc) return;
    const int t = i / N, n = i % N;
    prev[i] = t == 0 ? 1 : cap[n * Tc + t - 1];
    tgt[i] = cap[n * Tc + t];
}

hipError_t launch_prep_caption(const int32_t* cap, int32_t* prev, int32_t* tgt, int N, int Tc, hipStream_t st)
{
    hipLaunchKernelGGL(prep_caption_kernel, dim3((N * Tc + 255) / 256), dim3(256), 0, st, cap, prev, tgt, N, Tc);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Host glue of the REINFORCE step on the device (decode_captions_masks, cider_evaluation.py:145-172, and the objective's
// coefficients, reinforcement_multisampling_tf_s2vt.py:641-646): what used to be ~20 tiny tensor-library launches per step.
//   caption_mask: mask[n][t] = 1 up to and including the first <eos> = 0 of row n; target_tm[t*N + n] = ids[n][t];
//                 *mask_sum = sum(mask) (ONE workgroup: no zeroing, no atomics, deterministic)
//   pg_coef:      coef_tm[t*N + n] = mask[n][t] * (rewards[n] - baseline[n]) * scale
//   step_scalars: *loss = sum(coef * nll) / *msum_local;  *gscale = 1 / *gsum_global;  *sumsq = 0
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void caption_mask_kernel(const int32_t* ids, int N, int Tc, float* mask, int32_t* target_tm, float* mask_sum, float* mask_sum_copy)
{
    __shared__ float sh[4];
    float local = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
        bool alive = true;
        for (int t = 0; t < Tc; ++t) {
            const int32_t w = ids[(size_t)n * Tc + t];
            const float mk = alive ? 1.0f : 0.0f;
            if (mask) mask[(size_t)n * Tc + t] = mk;
            if (target_tm) target_tm[(size_t)t * N + n] = w;
            local += mk;
            if (w == 0) alive = false;
        }
    }
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float total = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        if (mask_sum) *mask_sum = total;
        if (mask_sum_copy) *mask_sum_copy = total;          // (the gradient bucket's tail slot: sum(mask) rides through the all-reduce there)
    }
}

hipError_t launch_caption_mask(const int32_t* ids, int N, int Tc, float* mask, int32_t* target_tm, float* mask_sum, float* mask_sum_copy, hipStream_t st)
{
    hipLaunchKernelGGL(caption_mask_kernel, dim3(1), dim3(256), 0, st, ids, N, Tc, mask, target_tm, mask_sum, mask_sum_copy);
    return hipGetLastError();
}

// ---- the coefficient / target preparation of the XE and the mixed (multitask) update as ONE launch each (round 4: the tensor
// library spent 8 and ~25 launches per step on them, 55 and 145 us of a 2.5 / 4.5 ms step).  One workgroup; the sums are over
// 0/1 masks (exact in any order), every other expression is written in the order the tensor-library code had it.
//   xe_prep (tf_s2vt.py:150-166): colsum[t] = sum_n mask[n][t];  coef_tm[t N + n] = q1 ? (colsum[t] * (1 / n_glob)) * loss_weight
//   : mask[n][t] * loss_weight;  target_tm[t N + n] = caption[n][t];  *msum = sum(mask)
__global__ __launch_bounds__(1024) void xe_prep_kernel(const float* mask, const int32_t* cap, int N, int Tc, float loss_weight, float n_glob,
                                                       int q1, float* coef_tm, int32_t* target_tm, float* msum)
{
    __shared__ float col[128];
    const int tid = threadIdx.x;
    if (tid < Tc) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += mask[(size_t)n * Tc + tid];
        col[tid] = s;
    }
    __syncthreads();
    if (tid == 0 && msum) {
        float s = 0.f;
        for (int t = 0; t < Tc; ++t) s += col[t];
        *msum = s;
    }
    const float inv_n = 1.0f / n_glob;           // (a tensor divided by a host scalar is a multiplication by its reciprocal in the tensor library: the same here)
    for (int i = tid; i < N * Tc; i += 1024) {
        const int t = i / N, n = i - t * N;
        coef_tm[i] = q1 ? (col[t] * inv_n) * loss_weight : mask[(size_t)n * Tc + t] * loss_weight;
        if (target_tm) target_tm[i] = cap[(size_t)n * Tc + t];
    }
}
hipError_t launch_xe_prep(const float* mask, const int32_t* cap, int N, int Tc, float loss_weight, float n_glob, int q1, float* coef_tm,
                          int32_t* target_tm, float* msum, hipStream_t st)
{
    if (N <= 0 || Tc <= 0 || Tc > 128) return hipErrorInvalidValue;
    hipLaunchKernelGGL(xe_prep_kernel, dim3(1), dim3(1024), 0, st, mask, cap, N, Tc, loss_weight, n_glob, q1, coef_tm, target_tm, msum);
    return hipGetLastError();
}

//   mixed_prep (reinforce_multitask_e2e_attribute_s2vt.py:850): rows 0 .. Ns-1 are the sampled captions, rows Ns .. Ns+B-1 the
//   ground truth.  coef_tm[t N + n] = n < Ns ? mask[n][t] * ((r[n] - b[n]) * one_minus_lam) / sum(mask)
//                                           : (q1 ? (colsum_gt[t] / n_glob_b) * loss_weight : gmask[.][t] * loss_weight) * (lam / sum(gmask));
//   smooth_tm = 0 | smoothing;  cap_all = [cap ; gcap], target_tm the same ids time-major;  sums = {sum(mask), sum(gmask)}
__global__ __launch_bounds__(1024) void mixed_prep_kernel(const float* mask, const float* gmask, const float* rewards, const float* baseline,
                                                          const int32_t* cap, const int32_t* gcap, int Ns, int B, int Tc, float one_minus_lam,
                                                          float lam, float loss_weight, int q1, float smoothing, float n_glob_b, float* coef_tm,
                                                          float* smooth_tm, int32_t* cap_all, int32_t* target_tm, float* sums)
{
    __shared__ float col[128], colp[128];
    __shared__ float s01[2];
    const int tid = threadIdx.x;
    if (tid < Tc) {
        float s = 0.f;
        for (int n = 0; n < B; ++n) s += gmask[(size_t)n * Tc + tid];
        col[tid] = s;
    } else if (tid >= 512 && tid - 512 < Tc) {
        const int t = tid - 512;
        float s = 0.f;
        for (int n = 0; n < Ns; ++n) s += mask[(size_t)n * Tc + t];
        colp[t] = s;
    }
    __syncthreads();
    if (tid == 0) {
        float a = 0.f, b = 0.f;
        for (int t = 0; t < Tc; ++t) { a += colp[t]; b += col[t]; }
        s01[0] = a; s01[1] = b;
        sums[0] = a; sums[1] = b;
    }
    __syncthreads();
    const float s0 = s01[0], xs = lam / s01[1], inv_nb = 1.0f / n_glob_b;
    const int N = Ns + B;
    for (int i = tid; i < N * Tc; i += 1024) {
        const int t = i / N, n = i - t * N;
        float c;
        if (n < Ns) {
            const float a = (rewards[n] - baseline[n]) * one_minus_lam;
            c = mask[(size_t)n * Tc + t] * a / s0;
        } else {
            const float x = q1 ? (col[t] * inv_nb) * loss_weight : gmask[(size_t)(n - Ns) * Tc + t] * loss_weight;
            c = x * xs;
        }
        coef_tm[i] = c;
        smooth_tm[i] = n < Ns ? 0.0f : smoothing;
        cap_all[i] = i < Ns * Tc ? cap[i] : gcap[i - Ns * Tc];
        target_tm[i] = n < Ns ? cap[(size_t)n * Tc + t] : gcap[(size_t)(n - Ns) * Tc + t];
    }
}
hipError_t launch_mixed_prep(const float* mask, const float* gmask, const float* rewards, const float* baseline, const int32_t* cap,
                             const int32_t* gcap, int Ns, int B, int Tc, float one_minus_lam, float lam, float loss_weight, int q1,
                             float smoothing, float n_glob_b, float* coef_tm, float* smooth_tm, int32_t* cap_all, int32_t* target_tm, float* sums, hipStream_t st)
{
    if (Ns <= 0 || B <= 0 || Tc <= 0 || Tc > 128) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mixed_prep_kernel, dim3(1), dim3(1024), 0, st, mask, gmask, rewards, baseline, cap, gcap, Ns, B, Tc, one_minus_lam, lam,
                       loss_weight, q1, smoothing, n_glob_b, coef_tm, smooth_tm, cap_all, target_tm, sums);
    return hipGetLastError();
}

//   mixed_loss: out = {sum over the sampled rows, sum over the ground-truth rows, their sum} of coef * nll; row of entry r =
//   (live_rows ? live_rows[r] : r) % N, sampled when < Ns
__global__ __launch_bounds__(1024) void mixed_loss_kernel(const float* coef, const float* nll, const int32_t* live_rows, int R, int N, int Ns,
                                                          float* out3)
{
    __shared__ float sh[2][32];
    float a = 0.f, b = 0.f;
    for (int r = threadIdx.x; r < R; r += 1024) {
        const int n = (live_rows ? live_rows[r] : r) % N;
        const float v = coef[r] * nll[r];
        if (n < Ns) a += v; else b += v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a; sh[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float x = 0.f, y = 0.f;
        for (int w = 0; w < 16; ++w) { x += sh[0][w]; y += sh[1][w]; }
        out3[0] = x; out3[1] = y; out3[2] = x + y;
    }
}
hipError_t launch_mixed_loss(const float* coef, const float* nll, const int32_t* live_rows, int R, int N, int Ns, float* out3, hipStream_t st)
{
    if (R < 0 || N <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mixed_loss_kernel, dim3(1), dim3(1024), 0, st, coef, nll, live_rows, R, N, Ns, out3);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void pg_coef_kernel(const float* mask, const float* rewards, const float* baseline, float scale, int N,
                                                      int Tc, float* coef_tm)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * Tc) return;
    const int t = i / N, n = i % N;
    const float adv = (rewards ? rewards[n] : 1.0f) - (baseline ? baseline[n] : 0.0f);
    coef_tm[i] = mask[(size_t)n * Tc + t] * (adv * scale);
}

hipError_t launch_pg_coef(const float* mask, const float* rewards, const float* baseline, float scale, int N, int Tc, float* coef_tm,
                          hipStream_t st)
{
    if (N * Tc <= 0) return hipSuccess;
    hipLaunchKernelGGL(pg_coef_kernel, dim3((N * Tc + 255) / 256), dim3(256), 0, st, mask, rewards, baseline, scale, N, Tc, coef_tm);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void step_scalars_kernel(const float* coef, const float* nll, int64_t R, const float* msum_local,
                                                           const float* gsum_global, float* loss, float* gscale, float* sumsq)
{
    __shared__ float sh[4];
    float local = 0.f;
    if (coef && nll)
        for (int64_t i = threadIdx.x; i < R; i += 256) local += coef[i] * nll[i];
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (loss) *loss = ((sh[0] + sh[1]) + (sh[2] + sh[3])) / (msum_local ? *msum_local : 1.0f);
        if (gscale) *gscale = 1.0f / (gsum_global ? *gsum_global : 1.0f);
        if (sumsq) *sumsq = 0.0f;
    }
}

hipError_t launch_step_scalars(const float* coef, const float* nll, int64_t R, const float* msum_local, const float* gsum_global, float* loss,
                               float* gscale, float* sumsq, hipStream_t st)
{
    hipLaunchKernelGGL(step_scalars_kernel, dim3(1), dim3(256), 0, st, coef, nll, R, msum_local, gsum_global, loss, gscale, sumsq);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// softmax / NLL rows, forward + backward in one pass over the logits.
//   lse = max + log(sum exp(l - max)); q = onehot*(1-s) + s/V; nll = -sum q*(l - lse)
//   dlogits = coef[row] * (softmax - q)      (written in place over the logits)
// XE (tf_s2vt.py:155-160): s = 0.05, coef from the Q1 batch-mean rule; PG
// (reinforcement_multisampling_tf_s2vt.py:286-291,643-646): s = 0, coef = (r-b)*mask.
// One workgroup per row; the row is re-read from L2 (48 KB at V=12k), HBM sees one read + one write.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_nll_kernel(float* logits, int ld, int V, const int32_t* target,
                                                          const float* coef, float smoothing_all, const float* smooth_rows,
                                                          float* nll, float* lp_t)
{
    const float smoothing = smooth_rows ? smooth_rows[blockIdx.x] : smoothing_all;      // per-row label smoothing (rows of two objectives in one pass)
    __shared__ float sh[8];
    const int row = blockIdx.x;
    float* l = logits + (size_t)row * ld;
    const int tid = threadIdx.x;
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(logits) & 15) == 0);
    const int V4 = vec ? (V >> 2) : 0;

    float mx = -INFINITY;
    for (int i = tid; i < V4; i += 256) {
        const float4 v = reinterpret_cast<const float4*>(l)[i];
        mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    for (int i = V4 * 4 + tid; i < V; i += 256) mx = fmaxf(mx, l[i]);
    mx = block_reduce<true>(mx, sh);

    float se = 0.f, sl = 0.f;   // sum exp(l-mx), sum l (for the smoothing term)
    for (int i = tid; i < V4; i += 256) {
        const float4 v = reinterpret_cast<const float4*>(l)[i];
        se += dm_expf(v.x - mx) + dm_expf(v.y - mx) + dm_expf(v.z - mx) + dm_expf(v.w - mx);
        sl += (v.x + v.y) + (v.z + v.w);
    }
    for (int i = V4 * 4 + tid; i < V; i += 256) { se += dm_expf(l[i] - mx); sl += l[i]; }
    se = block_reduce<false>(se, sh);
    sl = block_reduce<false>(sl, sh);
    const float lse = mx + dm_logf(se);
    const int tg = target[row];
    const float lt = l[tg];
    const float qoff = smoothing / (float)V;
    if (tid == 0) {
        // -sum q*(l-lse) = -(1-s)*(l_t - lse) - (s/V) * (sum l - V*lse)
        const float v = -(1.0f - smoothing) * (lt - lse) - qoff * (sl - (float)V * lse);
        if (nll) nll[row] = v;
        if (lp_t) lp_t[row] = lt - lse;
    }
    const float cf = coef[row];
    const float inv = 1.0f / se;
    __syncthreads();   // everyone has read l[tg] before it is overwritten
    for (int i = tid; i < V4; i += 256) {
        float4 v = reinterpret_cast<const float4*>(l)[i];
        v.x = cf * (dm_expf(v.x - mx) * inv - qoff);
        v.y = cf * (dm_expf(v.y - mx) * inv - qoff);
        v.z = cf * (dm_expf(v.z - mx) * inv - qoff);
        v.w = cf * (dm_expf(v.w - mx) * inv - qoff);
        const int b = i * 4;
        if (tg >= b && tg < b + 4) {
            const float d = cf * (1.0f - smoothing);
            if (tg == b) v.x -= d; else if (tg == b + 1) v.y -= d; else if (tg == b + 2) v.z -= d; else v.w -= d;
        }
        reinterpret_cast<float4*>(l)[i] = v;
    }
    for (int i = V4 * 4 + tid; i < V; i += 256) {
        float v = cf * (dm_expf(l[i] - mx) * inv - qoff);
        if (i == tg) v -= cf * (1.0f - smoothing);
        l[i] = v;
    }
}

// The same arithmetic, in the same per-thread order, with the row held in REGISTERS: thread t owns the float4s
// t, t+256, ... (NV4 of them, V <= 1024*NV4), loads them once, keeps exp(l - max) from the sum pass for the gradient
// pass -- HBM and L2 see one read and one write of the row, and each element costs one exp instead of two.
template <int NV4>
__global__ __launch_bounds__(256) void softmax_nll_reg_kernel(float* logits, int ld, int V, const int32_t* target,
                                                              const float* coef, float smoothing_all, const float* smooth_rows,
                                                              float* nll, float* lp_t)
{
    const float smoothing = smooth_rows ? smooth_rows[blockIdx.x] : smoothing_all;
    __shared__ float sh[8];
    const int row = blockIdx.x;
    float* l = logits + (size_t)row * ld;
    const int tid = threadIdx.x;
    const int V4 = V >> 2;                        // (V % 4 == 0 on this path)
    float4 x[NV4];
    float mx = -INFINITY;
#pragma unroll
    for (int n = 0; n < NV4; ++n) {
        const int i = tid + n * 256;
        if (i < V4) {
            x[n] = reinterpret_cast<const float4*>(l)[i];
            mx = fmaxf(fmaxf(mx, fmaxf(x[n].x, x[n].y)), fmaxf(x[n].z, x[n].w));
        }
    }
    mx = block_reduce<true>(mx, sh);
    const int tg = target[row];
    float lt_part = 0.f;                          // the owner of the target element contributes l[tg]
    float se = 0.f, sl = 0.f;
#pragma unroll
    for (int n = 0; n < NV4; ++n) {
        const int i = tid + n * 256;
        if (i < V4) {
            const float4 v = x[n];
            const int b = i * 4;
            if (tg >= b && tg < b + 4) lt_part = tg == b ? v.x : tg == b + 1 ? v.y : tg == b + 2 ? v.z : v.w;
            sl += (v.x + v.y) + (v.z + v.w);
            x[n].x = dm_expf(v.x - mx); x[n].y = dm_expf(v.y - mx); x[n].z = dm_expf(v.z - mx); x[n].w = dm_expf(v.w - mx);
            se += x[n].x + x[n].y + x[n].z + x[n].w;
        }
    }
    se = block_reduce<false>(se, sh);
    sl = block_reduce<false>(sl, sh);
    const float lt = block_reduce<false>(lt_part, sh);           // one non-zero term: exact
    const float lse = mx + dm_logf(se);
    const float qoff = smoothing / (float)V;
    if (tid == 0) {
        const float v = -(1.0f - smoothing) * (lt - lse) - qoff * (sl - (float)V * lse);
        if (nll) nll[row] = v;
        if (lp_t) lp_t[row] = lt - lse;
    }
    const float cf = coef[row];
    const float inv = 1.0f / se;
#pragma unroll
    for (int n = 0; n < NV4; ++n) {
        const int i = tid + n * 256;
        if (i < V4) {
            float4 v;
            v.x = cf * (x[n].x * inv - qoff);
            v.y = cf * (x[n].y * inv - qoff);
            v.z = cf * (x[n].z * inv - qoff);
            v.w = cf * (x[n].w * inv - qoff);
            const int b = i * 4;
            if (tg >= b && tg < b + 4) {
                const float d = cf * (1.0f - smoothing);
                if (tg == b) v.x -= d; else if (tg == b + 1) v.y -= d; else if (tg == b + 2) v.z -= d; else v.w -= d;
            }
            reinterpret_cast<float4*>(l)[i] = v;
        }
    }
}

hipError_t launch_softmax_nll(float* logits, int ld, int R, int V, const int32_t* target, const float* coef,
                              float smoothing, float* nll, float* lp_t, hipStream_t st, const float* smooth_rows)
{
    if (R <= 0) return hipSuccess;
    const bool vec = (ld & 3) == 0 && (V & 3) == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0;
    if (vec && V <= 1024 * 4)
        hipLaunchKernelGGL(softmax_nll_reg_kernel<4>, dim3(R), dim3(256), 0, st, logits, ld, V, target, coef, smoothing, smooth_rows, nll, lp_t);
    else if (vec && V <= 1024 * 12)
        hipLaunchKernelGGL(softmax_nll_reg_kernel<12>, dim3(R), dim3(256), 0, st, logits, ld, V, target, coef, smoothing, smooth_rows, nll, lp_t);
    else
        hipLaunchKernelGGL(softmax_nll_kernel, dim3(R), dim3(256), 0, st, logits, ld, V, target, coef, smoothing, smooth_rows, nll, lp_t);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// build_generator's word choice AS THE REFERENCE WRITES IT (tf_s2vt.py:208-209, the quirk switch of SURVEY A9):
//   p = exp(l) / sum_n exp(l[n])   -- no max shift, fp32 --   then argmax(p), first maximum wins, NaN never wins.
// A logit >= 88.72 overflows exp to +inf, the sum to +inf, that entry to inf/inf = NaN and every other to 0: the
// result is index 0 (<eos>), not argmax(l).  Summation order = the numeric contract's ("reductions that
// decide a token"): thread t adds elements t, t+256, ... ascending; xor-butterfly 32..1 inside each wave; then
// ((w0 + w1) + w2) + w3.  One workgroup per row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_unshifted_argmax_kernel(const float* logits, int ld, int V, int32_t* ids,
                                                                       float* probs)
{
    __shared__ float shs[4];
    __shared__ unsigned long long shk[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* l = logits + (size_t)row * ld;
    float part = 0.f;
    for (int i = tid; i < V; i += 256) part = part + dm_expf_ieee(l[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part = part + __shfl_xor(part, o, 64);
    if ((tid & 63) == 0) shs[tid >> 6] = part;
    __syncthreads();
    const float total = ((shs[0] + shs[1]) + shs[2]) + shs[3];
    // argmax over p >= 0 (or NaN): key = (bits of p) << 32 | ~index orders by value, then by LOWER index; NaN -> no key
    unsigned long long best = 0ull;
    for (int i = tid; i < V; i += 256) {
        const float pv = dm_expf_ieee(l[i]) / total;
        if (probs) probs[(size_t)row * V + i] = pv;
        if (pv == pv) {
            const unsigned long long k = ((unsigned long long)(__float_as_uint(pv + 0.0f) + 1u) << 32) | (uint32_t)(~(uint32_t)i);
            best = k > best ? k : best;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ob = __shfl_xor(best, o, 64);
        best = ob > best ? ob : best;
    }
    if ((tid & 63) == 0) shk[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        unsigned long long b = shk[0];
        for (int w = 1; w < 4; ++w) b = shk[w] > b ? shk[w] : b;
        ids[row] = b == 0ull ? 0 : (int32_t)(~(uint32_t)b);     // every entry NaN: the reducer keeps its initial index 0
    }
}

hipError_t launch_softmax_unshifted_argmax(const float* logits, int ld, int R, int V, int32_t* ids, float* probs, hipStream_t st)
{
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(softmax_unshifted_argmax_kernel, dim3(R), dim3(256), 0, st, logits, ld, V, ids, probs);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// BasicLSTMCell pointwise backward for one step (inverse of the EPI_LSTM epilogue):
//   dh   = sum_s dh_rec[s] (split-K slabs of dz_{t+1} @ Whh^T) + dout_ext * keepmask / keep
//   do~  = dh*tanh(c)*so*(1-so);  dc = dc_next + dh*so*(1-tanh(c)^2)
//   di~  = dc*tj*si*(1-si); dj~ = dc*si*(1-tj^2); df~ = dc*c_prev*sf*(1-sf); dc_prev = dc*sf
// gates = [si | tj | sf | so] saved by the forward.  dout_ext may be NULL (encode-stage LSTM2
// output is discarded, tf_s2vt.py:122).
// ---------------------------------------------------------------------------------------------
struct LstmBwdArgs {
    const float* gates; const float* c_new; const float* c_prev;   // c_prev == nullptr -> zeros (t = 0)
    const float* dh_rec; int nslab; size_t slab_stride;             // nullptr at the last step
    const float* dout_ext; int ld_ext;                              // gradient w.r.t. the DROPPED output
    float* dc;                                                      // in: dc_next (or nullptr at last step) / out: dc_prev
    const float* dc_in;
    float* dz;                                                      // [M, 4H]
    int M, H;
    float keep; uint32_t seed_lo, seed_hi, drop_code;
    const int32_t* video_id; const int32_t* sample_id;
};

__global__ __launch_bounds__(256) void lstm_bwd_pointwise_kernel(const LstmBwdArgs a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.M * a.H) return;
    const int m = i / a.H, u = i % a.H;
    const int H = a.H;
    float dh = 0.f;
    if (a.dh_rec)
        for (int s = 0; s < a.nslab; ++s) dh += a.dh_rec[(size_t)s * a.slab_stride + i];
    if (a.dout_ext) {
        float d = a.dout_ext[(size_t)m * a.ld_ext + u];
        if (a.keep < 1.0f)
            d = (d / a.keep) * dropout_keep01(a.seed_lo, a.seed_hi, (uint32_t)a.video_id[m], (uint32_t)a.sample_id[m],
                                              a.drop_code, (uint32_t)u, a.keep);
        dh += d;
    }
    const float* g = a.gates + (size_t)m * 4 * H + u;
    const float si = g[0], tj = g[H], sf = g[2 * H], so = g[3 * H];
    const float tc = dm_tanhf(a.c_new[i]);
    const float cp = a.c_prev ? a.c_prev[i] : 0.f;
    float dc = dh * so * (1.f - tc * tc);
    if (a.dc_in) dc += a.dc_in[i];
    float* z = a.dz + (size_t)m * 4 * H + u;
    z[0] = dc * tj * si * (1.f - si);
    z[H] = dc * si * (1.f - tj * tj);
    z[2 * H] = dc * cp * sf * (1.f - sf);
    z[3 * H] = dh * tc * so * (1.f - so);
    a.dc[i] = dc * sf;
}

hipError_t launch_lstm_bwd_pointwise(const float* gates, const float* c_new, const float* c_prev, const float* dh_rec,
                                     int nslab, size_t slab_stride, const float* dout_ext, int ld_ext, const float* dc_in,
                                     float* dc_out, float* dz, int M, int H, float keep, uint64_t seed, uint32_t drop_code,
                                     const int32_t* video_id, const int32_t* sample_id, hipStream_t st)
{
    LstmBwdArgs a;
    a.gates = gates; a.c_new = c_new; a.c_prev = c_prev; a.dh_rec = dh_rec; a.nslab = nslab; a.slab_stride = slab_stride;
    a.dout_ext = dout_ext; a.ld_ext = ld_ext; a.dc = dc_out; a.dc_in = dc_in; a.dz = dz; a.M = M; a.H = H; a.keep = keep;
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.drop_code = drop_code;
    a.video_id = video_id; a.sample_id = sample_id;
    hipLaunchKernelGGL(lstm_bwd_pointwise_kernel, dim3((M * H + 255) / 256), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// DropoutWrapper output of a cell whose clean state trajectory is shared by the rep sample rows of a
// video (LSTM1 in build_loss sees the same frames in every one of the K tiled copies,
// reinforcement_multisampling_tf_s2vt.py:779-782, so its state is computed once per video), and the
// adjoint reduction.  Same expression as the EPI_LSTM epilogue: (h / keep) * floor(keep + u).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void expand_dropout_kernel(const float* h, float* out, int T, int B, int N, int H,
                                                             float keep, uint32_t seed_lo, uint32_t seed_hi,
                                                             uint32_t code_base, const int32_t* video_id,
                                                             const int32_t* sample_id)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)T * N * H) return;
    const int u = (int)(i % H);
    const size_t tn = i / H;
    const int n = (int)(tn % N), t = (int)(tn / N);
    float v = h[((size_t)t * B + n % B) * H + u];
    if (keep < 1.0f)
        v = (v / keep) * dropout_keep01(seed_lo, seed_hi, (uint32_t)video_id[n], (uint32_t)sample_id[n], code_base + (uint32_t)t,
                                        (uint32_t)u, keep);
    out[i] = v;
}

hipError_t launch_expand_dropout(const float* h, float* out, int T, int B, int N, int H, float keep, uint64_t seed,
                                 uint32_t code_base, const int32_t* video_id, const int32_t* sample_id, hipStream_t st)
{
    const size_t n = (size_t)T * N * H;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(expand_dropout_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, h, out, T, B, N, H, keep,
                       (uint32_t)seed, (uint32_t)(seed >> 32), code_base, video_id, sample_id);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void reduce_dropout_kernel(const float* dout, int ld_out, float* dh, int T, int B, int N,
                                                             int H, float keep, uint32_t seed_lo, uint32_t seed_hi,
                                                             uint32_t code_base, const int32_t* video_id,
                                                             const int32_t* sample_id)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)T * B * H) return;
    const int u = (int)(i % H);
    const size_t tj = i / H;
    const int j = (int)(tj % B), t = (int)(tj / B);
    float acc = 0.f;
    for (int n = j; n < N; n += B) {
        float d = dout[((size_t)t * N + n) * ld_out + u];
        if (keep < 1.0f)
            d = (d / keep) * dropout_keep01(seed_lo, seed_hi, (uint32_t)video_id[n], (uint32_t)sample_id[n],
                                            code_base + (uint32_t)t, (uint32_t)u, keep);
        acc += d;
    }
    dh[i] = acc;
}

hipError_t launch_reduce_dropout(const float* dout, int ld_out, float* dh, int T, int B, int N, int H, float keep,
                                 uint64_t seed, uint32_t code_base, const int32_t* video_id, const int32_t* sample_id,
                                 hipStream_t st)
{
    const size_t n = (size_t)T * B * H;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(reduce_dropout_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dout, ld_out, dh, T, B, N, H,
                       keep, (uint32_t)seed, (uint32_t)(seed >> 32), code_base, video_id, sample_id);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// out[n] += sum_m X[m, n]      (bias gradients).  64 columns x 256-row slabs per workgroup.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float* X, int ld, int M, int N, float* out)
{
    __shared__ float sh[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int r0 = blockIdx.y * 256 + (threadIdx.x >> 6);
    float s = 0.f;
    if (c < N)
        for (int r = r0; r < M && r < (int)(blockIdx.y + 1) * 256; r += 4) s += X[(size_t)r * ld + c];
    sh[threadIdx.x >> 6][threadIdx.x & 63] = s;
    __syncthreads();
    if (threadIdx.x < 64 && c < N) atomicAdd(out + c, (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]));
}

hipError_t launch_colsum(const float* X, int ld, int M, int N, float* out, hipStream_t st)
{
    if (M <= 0 || N <= 0) return hipSuccess;
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, (M + 255) / 256), dim3(256), 0, st, X, ld, M, N, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// dst[i] = sum_s slabs[s * stride + i]   (split-K partial products of an order-free contraction; dst may be slab 0)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sum_slabs_kernel(float* dst, const float* slabs, int nslab, size_t stride, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 acc = reinterpret_cast<const f32x4*>(slabs)[i];
    for (int s = 1; s < nslab; ++s) {
        const f32x4 v = reinterpret_cast<const f32x4*>(slabs + (size_t)s * stride)[i];
        acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    reinterpret_cast<f32x4*>(dst)[i] = acc;
}

// dst[r, :] = src[idx[r], :] (rows of 16-byte multiples) and dst[r] = src[idx[r]]: the packed copies the live-row backward works on
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* src, int ld, const int32_t* idx, int R, int C4, float* dst, int ldd)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float4* s4 = reinterpret_cast<const float4*>(src + (size_t)idx[r] * ld);
    float4* d4 = reinterpret_cast<float4*>(dst + (size_t)r * ldd);
    for (int c = threadIdx.x & 63; c < C4; c += 64) d4[c] = s4[c];
}
// dst[idx[r], :] = src[r, :]: packed rows back to their places (distinct indices)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* src, int ld, const int32_t* idx, int R, int C4, float* dst, int ldd)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float4* s4 = reinterpret_cast<const float4*>(src + (size_t)r * ld);
    float4* d4 = reinterpret_cast<float4*>(dst + (size_t)idx[r] * ldd);
    for (int c = threadIdx.x & 63; c < C4; c += 64) d4[c] = s4[c];
}
hipError_t launch_scatter_rows(const float* src, int ld, const int32_t* idx, int R, int C, float* dst, int ldd, hipStream_t st)
{
    if (R <= 0) return hipSuccess;
    if ((C & 3) || (ld & 3) || (ldd & 3) || (reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 15)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, st, src, ld, idx, R, C / 4, dst, ldd);
    return hipGetLastError();
}
__global__ void gather_i32_kernel(const int32_t* src, const int32_t* idx, int R, int32_t* dst)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R) dst[r] = src[idx[r]];
}
hipError_t launch_gather_rows(const float* src, int ld, const int32_t* idx, int R, int C, float* dst, int ldd, hipStream_t st)
{
    if (R <= 0) return hipSuccess;
    if ((C & 3) || (ld & 3) || (ldd & 3) || (reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 15)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, st, src, ld, idx, R, C / 4, dst, ldd);
    return hipGetLastError();
}
// ---------------------------------------------------------------------------------------------
// Row order of a live-row unroll for the recurrence kernels (chain.hip, lstm_chain4_kernel<.., true>): from the list of
// unmasked (step, row) pairs t * N + n -- each row's live steps a PREFIX 0 .. len-1 of the decode steps, the precondition of
// the *_live entry points -- the rows sorted by length, longest first (stable: equal lengths keep their order), and the number
// of rows still live at every step of the recurrence (every row during the Tv encode steps).  One workgroup, N <= 1024.
// ---------------------------------------------------------------------------------------------
// The list is VALIDATED here (the *_live contract: time-major indices, strictly ascending, every row's live steps a PREFIX of the
// decode steps -- what a mask up to a first <eos> gives; model.live_rows() checks the same on the host): len[n] entries of row n
// must be steps 0 .. len[n]-1.  A list with a hole, a duplicate or an index out of range would make LSTM2 stop a row too early and
// the gathered gradient products miss rows -- silently.  So: the recurrences get the DENSE order (perm = identity, every row live at
// every step: nothing stops early), and the sticky fault of chain_common.h is raised (device word: queued Adam launches skip their
// update; host-mapped counter: every later entry point returns S2VT_E_CHAIN_TIMEOUT until s2vt_chain_ack, model.check_health()
// raises); word 1 of the fault buffer says why (1 = invalid live list).
__global__ __launch_bounds__(1024) void row_order_kernel(const int32_t* live_rows, int n_live, int N, int Tv, int Tc, int32_t* perm, int32_t* nlive,
                                                         unsigned* fault, unsigned* status)
{
    __shared__ int len[1024];
    __shared__ int maxt[1024];
    __shared__ int cnt[128];
    __shared__ int bad;
    const int tid = threadIdx.x;
    len[tid] = 0;
    maxt[tid] = -1;
    if (tid < 128) cnt[tid] = 0;
    if (tid == 0) bad = 0;
    __syncthreads();
    for (int r = tid; r < n_live; r += 1024) {
        const int idx = live_rows[r];
        const int t = idx / N, n = idx - t * N;
        if (idx < 0 || t >= Tc || t >= 128 || (r > 0 && live_rows[r - 1] >= idx)) { bad = 1; continue; }
        atomicAdd(&len[n], 1); atomicAdd(&cnt[t], 1); atomicMax(&maxt[n], t);
    }
    __syncthreads();
    if (tid < N && len[tid] != maxt[tid] + 1) bad = 1;
    __syncthreads();
    const bool invalid = bad != 0;
    if (tid < N) {
        const int mine = len[tid];
        int rank = 0;
        for (int m = 0; m < N; ++m) {
            const int o = len[m];
            rank += (o > mine || (o == mine && m < tid)) ? 1 : 0;
        }
        if (invalid) perm[tid] = tid; else perm[rank] = tid;
    }
    for (int t = tid; t < Tv + Tc; t += 1024) nlive[t] = (t < Tv || invalid) ? N : cnt[t - Tv];
    if (invalid && tid == 0) {
        if (fault) {
            __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(fault + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (status) __hip_atomic_fetch_add(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
hipError_t launch_row_order(const int32_t* live_rows, int n_live, int N, int Tv, int Tc, int32_t* perm, int32_t* nlive, hipStream_t st)
{
    if (!live_rows || n_live < 0 || N <= 0 || N > 1024 || Tv < 0 || Tc <= 0 || Tc > 128 || !perm || !nlive) return hipErrorInvalidValue;
    ChainHost h;
    const bool have = chain_host(&h);              // (no persistent-launch state on this device: nothing to raise the fault on; the order is still made dense)
    hipLaunchKernelGGL(row_order_kernel, dim3(1), dim3(1024), 0, st, live_rows, n_live, N, Tv, Tc, perm, nlive, have ? h.fault : nullptr,
                       have ? h.status_dev : nullptr);
    return hipGetLastError();
}

hipError_t launch_gather_i32(const int32_t* src, const int32_t* idx, int R, int32_t* dst, hipStream_t st)
{
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_i32_kernel, dim3((R + 255) / 256), dim3(256), 0, st, src, idx, R, dst);
    return hipGetLastError();
}

// Several buffers that must read as zeros when a pass starts (initial states, hand-off counters, fragment images), in ONE
// launch: each hipMemsetAsync is a ~5 us kernel of its own, and a REINFORCE step had 19 of them.
__global__ void zero_regions_kernel(const ZeroList z)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x, first = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int r = 0; r < z.count; ++r) {
        uint32_t* p = z.p[r];
        const size_t n = z.n[r];
        if ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
            uint4* p4 = reinterpret_cast<uint4*>(p);
            const size_t n4 = n >> 2;
            for (size_t i = first; i < n4; i += stride) p4[i] = make_uint4(0u, 0u, 0u, 0u);
            for (size_t i = (n4 << 2) + first; i < n; i += stride) p[i] = 0u;
        } else {
            for (size_t i = first; i < n; i += stride) p[i] = 0u;
        }
    }
}

hipError_t launch_zero_regions(const ZeroList& z, hipStream_t st)
{
    size_t most = 0;
    for (int r = 0; r < z.count; ++r) most = z.n[r] > most ? z.n[r] : most;
    if (z.count <= 0 || most == 0) return hipSuccess;
    const size_t want = (most / 4 + 255) / 256;
    const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 1024 ? 1024 : want));
    hipLaunchKernelGGL(zero_regions_kernel, dim3(blocks), dim3(256), 0, st, z);
    return hipGetLastError();
}

// Several device-to-device copies in ONE launch of a library kernel (the teacher-forced pass taking LSTM1's trajectory from the
// sampler pass was three hipMemcpyAsync = three runtime blit kernels per step).  16-byte aligned regions, sizes in 16-byte words.
__global__ __launch_bounds__(256) void copy_regions_kernel(const CopyList c)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x, first = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int r = 0; r < c.count; ++r) {
        const uint4* __restrict__ s = c.src[r];
        uint4* __restrict__ d = c.dst[r];
        const size_t n = c.n16[r];
        for (size_t i = first; i < n; i += stride) d[i] = s[i];
    }
}

hipError_t launch_copy_regions(const CopyList& c, hipStream_t st)
{
    size_t most = 0;
    for (int r = 0; r < c.count; ++r) most = c.n16[r] > most ? c.n16[r] : most;
    if (c.count <= 0 || most == 0) return hipSuccess;
    const size_t want = (most + 255) / 256;
    const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
    hipLaunchKernelGGL(copy_regions_kernel, dim3(blocks), dim3(256), 0, st, c);
    return hipGetLastError();
}

hipError_t launch_sum_slabs(float* dst, const float* slabs, int nslab, size_t stride, size_t n, hipStream_t st)
{
    if (n == 0 || nslab <= 0) return hipSuccess;
    if ((n & 3) || (stride & 3)) return hipErrorInvalidValue;
    const size_t n4 = n >> 2;
    hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, dst, slabs, nslab, stride, n4);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// dWemb[idx[r], :] += dE[r, :]   (gradient of tf.nn.embedding_lookup; one wave per row,
// 256 contiguous bytes per atomic wave-instruction).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* dE, int ld, const int32_t* idx, int R, int E,
                                                               float* dW, int ldw)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int row = idx[r];
    for (int e = threadIdx.x & 63; e < E; e += 64) atomicAdd(dW + (size_t)row * ldw + e, dE[(size_t)r * ld + e]);
}

hipError_t launch_scatter_add_rows(const float* dE, int ld, const int32_t* idx, int R, int E, float* dW, int ldw,
                                   hipStream_t st)
{
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, st, dE, ld, idx, R, E, dW, ldw);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// out[c, r] = in[r, c]   (32x32 tiles through LDS, padded rows)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* in, int ldi, float* out, int ldo, int R, int Cc)
{
    __shared__ float t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < R && c0 + tx < Cc) t[j][tx] = in[(size_t)(r0 + j) * ldi + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < Cc && r0 + tx < R) out[(size_t)(c0 + j) * ldo + r0 + tx] = t[tx][j];
}

hipError_t launch_transpose(const float* in, int ldi, float* out, int ldo, int R, int Cc, hipStream_t st)
{
    if (R <= 0 || Cc <= 0) return hipSuccess;
    hipLaunchKernelGGL(transpose_kernel, dim3((Cc + 31) / 32, (R + 31) / 32), dim3(256), 0, st, in, ldi, out, ldo, R, Cc);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// optimizer (reinforcement_multisampling_tf_s2vt.py:638-652)
//   finalize: g = g * (*gscale) + wd * theta  (per-segment wd), accumulate sum g^2 -> *sumsq
//   adam    : s = clip / max(sqrt(*sumsq), clip);  g' = g*s;  m,v update;  theta -= lr_t*m/(sqrt(v)+eps)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grad_finalize_kernel(float* g, const float* theta, int64_t n, const float* gscale,
                                                            float wd, float* sumsq)
{
    __shared__ float sh[8];
    const float sc = gscale ? *gscale : 1.0f;
    float acc = 0.f;
    const int64_t stride = (int64_t)gridDim.x * 256;
    // 16-byte pieces where the segment allows (3.7 -> ~5 TB/s: the scalar form left a quarter of the HBM rate unused); per element the
    // same expressions
    const bool vec = ((reinterpret_cast<uintptr_t>(g) | (wd != 0.f ? reinterpret_cast<uintptr_t>(theta) : 0)) & 15u) == 0;
    const int64_t n4 = vec ? n >> 2 : 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 v = reinterpret_cast<const float4*>(g)[i];
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        if (wd != 0.f) {
            const float4 th = reinterpret_cast<const float4*>(theta)[i];
            v.x += wd * th.x; v.y += wd * th.y; v.z += wd * th.z; v.w += wd * th.w;
        }
        reinterpret_cast<float4*>(g)[i] = v;
        acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float v = g[i] * sc;
        if (wd != 0.f) v += wd * theta[i];
        g[i] = v;
        acc += v * v;
    }
    acc = block_reduce<false>(acc, sh);
    if (threadIdx.x == 0) atomicAdd(sumsq, acc);
}

hipError_t launch_grad_finalize(float* g, const float* theta, int64_t n, const float* gscale, float wd, float* sumsq,
                                hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(grad_finalize_kernel, dim3(blocks), dim3(256), 0, st, g, theta, n, gscale, wd, sumsq);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void adam_tf_kernel(float* theta, const float* g, float* m, float* v, int64_t n,
                                                      const float* sumsq, float clip, float lr_t, float b1, float b2,
                                                      float eps, const unsigned* fault, int32_t* applied_step, int32_t step)
{
    if (fault && *fault != 0u) return;             // a persistent recurrence upstream timed out: leave the variables alone
    if (applied_step && blockIdx.x == 0 && threadIdx.x == 0) *applied_step = step;
    float s = 1.0f;
    if (sumsq && clip > 0.f) {
        const float nrm = sqrtf(*sumsq);
        s = clip / fmaxf(nrm, clip);
    }
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float gi = g[i] * s;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        theta[i] = theta[i] - lr_t * mi / (sqrtf(vi) + eps);
    }
}

hipError_t launch_adam_tf(float* theta, const float* g, float* m, float* v, int64_t n, const float* sumsq, float clip,
                          float lr_t, float b1, float b2, float eps, hipStream_t st, const unsigned* fault, int32_t* applied_step,
                          int32_t step)
{
    if (n <= 0) return hipSuccess;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_tf_kernel, dim3(blocks), dim3(256), 0, st, theta, g, m, v, n, sumsq, clip, lr_t, b1, b2, eps, fault,
                       applied_step, step);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// TN GEMM launcher
// ---------------------------------------------------------------------------------------------
namespace {
typedef void (*TnFn)(const TnKArgs);
struct TnCfg { TnFn vec, scalar; int BMo, BNo, NT, lds; };

template <int WM, int WN, int TM, int TN>
constexpr TnCfg tn_entry()
{
    constexpr int BMo = WM * TM * 16, BNo = WN * TN * 16;
    constexpr int SAo = (BMo % 32 == 16) ? BMo : BMo + 16, SBo = (BNo % 32 == 16) ? BNo : BNo + 16;
    return TnCfg{gemm_tn_kernel<WM, WN, TM, TN, true>, gemm_tn_kernel<WM, WN, TM, TN, false>, BMo, BNo, 64 * WM * WN,
                 2 * 32 * (SAo + SBo) * 4};
}
const TnCfg kTn[] = {tn_entry<2, 2, 4, 4>() /*128x128*/, tn_entry<2, 2, 2, 4>() /*64x128*/, tn_entry<2, 2, 2, 2>() /*64x64*/,
                     tn_entry<2, 2, 3, 3>() /*96x96*/, tn_entry<2, 2, 2, 3>() /*64x96*/, tn_entry<2, 2, 3, 4>() /*96x128*/};
struct TnDma { TnFn plain, gather; int lds, br, wgs_per_cu; };
template <int BR, int NB>
constexpr TnDma tn_dma_entry()
{
    constexpr int lds = NB * BR * 256 * 4, by_lds = 160 * 1024 / lds;
    return TnDma{gemm_tn_dma_kernel<false, BR, NB>, gemm_tn_dma_kernel<true, BR, NB>, lds, BR, by_lds < 3 ? by_lds : 3};   // 136 VGPRs: three waves per SIMD
}
// LDS-DMA forms of the 128x128 tile: {rows per chunk, LDS buffers}; [0] is the one used (the rest: S2VT_TN_DMA=2.. dev knob)
const TnDma kTnDma[] = {tn_dma_entry<16, 2>(), tn_dma_entry<32, 2>()};
std::once_flag g_tn_once;
hipError_t g_tn_attr_err = hipSuccess;
}  // namespace

hipError_t launch_gemm_tn(const TnArgs& a, hipStream_t st)
{
    std::call_once(g_tn_once, [] {
        for (const TnDma& d : kTnDma)
            for (TnFn fn : {d.plain, d.gather}) {
                const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, d.lds);
                if (e != hipSuccess && g_tn_attr_err == hipSuccess) g_tn_attr_err = e;
            }
        for (const TnCfg& c : kTn)
            for (TnFn fn : {c.vec, c.scalar}) {
                const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, c.lds);
                if (e != hipSuccess && g_tn_attr_err == hipSuccess) g_tn_attr_err = e;
            }
    });
    if (g_tn_attr_err != hipSuccess) return g_tn_attr_err;
    if (a.Mred <= 0 || a.Kout <= 0 || a.N <= 0) return hipSuccess;
    auto tiles = [&](const TnCfg& c) { return (long)((a.Kout + c.BMo - 1) / c.BMo) * ((a.N + c.BNo - 1) / c.BNo); };
    const bool vec = ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.B)) & 15) == 0 && (a.lda & 3) == 0 &&
                     (a.ldb & 3) == 0 && (a.Kout & 3) == 0 && (a.N & 3) == 0 &&
                     // the vector path addresses a chunk (32 rows) of each operand by 32-bit byte offsets from a base it
                     // re-computes per chunk; a gathered A must fit a 2 GiB window (its extent is the caller's table)
                     (size_t)32 * a.lda * 4 < (1ull << 31) && (size_t)32 * a.ldb * 4 < (1ull << 31) &&
                     !(a.rowidx && a.gather_rows > 0 && (size_t)a.gather_rows * a.lda * 4 >= (1ull << 31));
    // the 128x128 vector path stages by LDS-DMA (rows must be 16-byte aligned, which `vec` already says)
    static const int use_dma = [] { const char* e = getenv("S2VT_TN_DMA"); return e ? atoi(e) : 1; }();       // dev knob
    const bool dma_on = vec && use_dma >= 1 && use_dma <= (int)(sizeof(kTnDma) / sizeof(kTnDma[0]));
    const int maxs = (a.Mred + 255) / 256;
    int ci = 0;
    // fewer than 200 tiles of 128x128: the smaller tiles -- unless the LDS-DMA tile can fill the chip with reduction slabs
    const bool dma_fills = dma_on && tiles(kTn[0]) * (maxs < 64 ? maxs : 64) >= 512 && a.Kout >= 96 && a.N >= 96;
    if (tiles(kTn[0]) < 200 && !dma_fills) ci = 1;
    if (ci == 1 && tiles(kTn[1]) < 200) ci = 2;
    static const int force = [] { const char* e = getenv("S2VT_TN_CFG"); return e ? atoi(e) : -1; }();       // dev knob
    if (force >= 0 && ci == 0) ci = force;
    const TnCfg& c = kTn[ci];
    const long nt = tiles(c);
    const bool dma = dma_on && ci == 0;
    const TnDma& dm = kTnDma[dma ? use_dma - 1 : 0];
    const TnFn fn = dma ? (a.rowidx ? dm.gather : dm.plain) : (vec ? c.vec : c.scalar);
    const int lds = dma ? dm.lds : c.lds;
    int splits = 1;
    static const int tn_target = [] { const char* e = getenv("S2VT_TN_WGS"); return e ? atoi(e) : 0; }();      // dev knob: workgroups wanted (the rule of the register-staged tiles)
    if (dma && tn_target == 0) {
        // reduction slabs by cost: rounds of co-resident workgroups x (chunks per workgroup + its fixed prologue / epilogue,
        // priced in chunks).  752 tiles of the vocabulary gradient are one round of 768 slots -- no split, no atomics.
        static const int ovh = [] { const char* e = getenv("S2VT_TN_OVH"); return e ? atoi(e) : 12; }();       // dev knob
        const long slots = 256L * dm.wgs_per_cu;
        long best = -1;
        for (int sp = 1; sp <= (maxs < 64 ? maxs : 64); ++sp) {
            const long rounds = (nt * sp + slots - 1) / slots;
            const long chunks = ((a.Mred + sp - 1) / sp + dm.br - 1) / dm.br;
            const long cost = rounds * (chunks + ovh) * 64 + sp;                  // ties: fewer slabs
            if (best < 0 || cost < best) { best = cost; splits = sp; }
        }
    } else {
        const int target = tn_target ? tn_target : 1024;                          // >= ~4 per CU keeps the MFMA pipes fed (measured 62 -> 88 TFLOP/s)
        if (nt < target) {
            splits = (int)((target + nt - 1) / nt);
            if (splits > maxs) splits = maxs;
            if (splits < 1) splits = 1;
        }
    }
    TnKArgs k;
    k.A = a.A; k.rowidx = a.rowidx; k.lda = a.lda; k.B = a.B; k.ldb = a.ldb; k.C = a.C; k.ldc = a.ldc;
    k.Mred = a.Mred; k.Kout = a.Kout; k.N = a.N;
    k.mper = ((a.Mred + splits - 1) / splits + 31) / 32 * 32;
    splits = (a.Mred + k.mper - 1) / k.mper;
    k.atomic = splits > 1 ? 1 : 0;
    k.splits = splits;
    static const int xmap = [] { const char* e = getenv("S2VT_TN_XMAP"); return e ? atoi(e) : 1; }();         // dev knob
    k.xmap = xmap;
    static const int prio = [] { const char* e = getenv("S2VT_TN_PRIO"); return e ? atoi(e) : 32; }();       // dev knob (tools/ab_tn_prio.sh)
    k.prio_rot = prio;
    const long nrow = (a.Kout + c.BMo - 1) / c.BMo, ncol = (a.N + c.BNo - 1) / c.BNo;
    const dim3 grid = xmap ? dim3((unsigned)(8 * nrow * ((ncol * splits + 7) / 8))) : dim3((unsigned)nt, (unsigned)splits);
    k.accumulate = a.accumulate;
    k.colsum = nullptr;
    if (k.atomic && !a.accumulate) {
        hipError_t e = hipMemset2DAsync(a.C, (size_t)a.ldc * 4, 0, (size_t)a.N * 4, a.Kout, st);
        if (e != hipSuccess) return e;
    }
    if (a.colsum) {
        if (vec) k.colsum = a.colsum;                                   // fused: the B tiles pass through registers anyway
        else {
            hipError_t e = launch_colsum(a.B, a.ldb, a.Mred, a.N, a.colsum, st);
            if (e != hipSuccess) return e;
        }
    }
    const int pcfg = dma ? 6 : ci;                                       // profiler row of the LDS-DMA tile
    if (!prof_wants(3, pcfg)) {
        hipLaunchKernelGGL(fn, grid, dim3(c.NT), lds, st, k);
        return hipGetLastError();
    }
    hipEvent_t e0, e1;
    hipError_t pe = prof_events(&e0, &e1);
    if (pe != hipSuccess) return pe;
    static const char* names[] = {"tn128x128(2x2)", "tn64x128(2x2)", "tn64x64(2x2)", "tn96x96(2x2)", "tn64x96(2x2)", "tn96x128(2x2)", "tn128x128(dma)"};
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL(fn, grid, dim3(c.NT), lds, st, k);
    (void)hipEventRecord(e1, st);
    prof_record(3, pcfg, names[pcfg], 2.0 * a.Mred * (double)a.Kout * a.N, e0, e1);
    return hipGetLastError();
}

}  // namespace s2vt
