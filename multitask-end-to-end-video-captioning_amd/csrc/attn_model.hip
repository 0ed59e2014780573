// attn_model.hip -- the temporal-attention captioner of original_attention.py as whole-model entry points:
// build_model's unroll with saved activations (:88-150), back-propagation through it (the train_op of train(), :436-441),
// and the greedy sampler of build_generator / build_sampler (:155-251).  All loops run inside the library; the caller
// passes device pointers and one workspace.
//
// Per decode step the recurrence is  query -> score/softmax/context -> LSTM3:
//     hWa_t  = out_{t-1} @ Wa                              (out = the DropoutWrapper output: `h_prev = output1`, :135)
//     a_t, ctx_t = attention(hWa_t, P, V)                  (attn.hip: one launch)
//     z_t    = emb_t @ W3[H:2H]  ->  h_{t-1} @ W3[2H:3H]  ->  ctx_t @ W3[0:H]   (+ b3;  ONE ascending-k chain, blocks in
//              order of availability -- the numeric contract)
// and everything that does not feed the recurrence is batched over all steps: the embedding rows of W3 (hoisted, one
// product for all steps: the chain's first block), the output layer tanh([emb ; ctx ; out] @ Wp + bp) and the vocabulary
// logits after the loop, every weight gradient and the [out | ctx | emb] data gradient of the output layer in the
// backward.  Forward activations are bit-identical to oracle/s2vt_oracle.py::attention_forward; gradients are order-free.
#include <hip/hip_runtime.h>

#include <cstring>

#include <mutex>

#include "api_util.h"
#include "detmath.h"

using namespace s2vt_api;

namespace {

constexpr int kXSlabs = 8;        // most split-K slabs of the per-step product dz @ W3^T  ([B, 3H], K = 4H)
constexpr int kQSlabs = 4;        // ... of dhWa @ Wa^T ([B, H], K = H)
constexpr uint32_t kDropCode3 = 768u;   // dropout stream of LSTM3: code = 768 + decode step (layer 3 * 256)

__global__ void attn_enc_index_kernel(int32_t* idx, int B, int Tv)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Tv) return;
    const int t = i / B, j = i % B;
    idx[i] = j * Tv + t;     // row of video[B*Tv, d] feeding time-major row (frame t, video j)   (the transpose of :98)
}

__global__ void attn_rows_kernel(int32_t* vid, int32_t* sid, int B, int video_base)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    vid[i] = video_base + i;
    sid[i] = -1;             // greedy: argmax of the logits, no noise
}

__global__ void attn_unpack_ids_kernel(const unsigned long long* packed, int32_t* ids, int R, int T, int stride)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * T) return;
    const int m = i / T, t = i % T;
    ids[i] = (int32_t)(~(uint32_t)packed[((size_t)t * R + m) * stride]);
}

__global__ __launch_bounds__(256) void attn_tanh_bwd_kernel(float* dy, const float* y, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 d = reinterpret_cast<float4*>(dy)[i];
    const float4 v = reinterpret_cast<const float4*>(y)[i];
    d.x *= 1.f - v.x * v.x; d.y *= 1.f - v.y * v.y; d.z *= 1.f - v.z * v.z; d.w *= 1.f - v.w * v.w;
    reinterpret_cast<float4*>(dy)[i] = d;
}

// BasicLSTMCell backward of one decode step of LSTM3.  The gradient w.r.t. the step's DROPPED output has two sources -- the
// output layer (dcat's first H columns) and the next step's attention query (split-K slabs of dhWa @ Wa^T) -- and goes back
// through the DropoutWrapper mask; the gradient w.r.t. the clean h comes from the next step's recurrent rows (the last H
// columns of the slabs of dz @ W3^T).
struct AttnCellBwdArgs {
    const float* gates; const float* c_new; const float* c_prev;
    const float* dcat; int ld_cat;                         // [B, 3H]: columns [0, H) = d(out)
    const float* dqs; int nq; size_t q_stride;             // slabs [nq][B][H] or NULL
    const float* dxs; int nx; size_t x_stride; int ld_x; int x_col0;   // slabs [nx][B][3H], recurrent block at x_col0, or NULL
    const float* dc_in; float* dc_out; float* dz;
    int M, H;
    float keep; uint32_t seed_lo, seed_hi, drop_code;
    const int32_t* video_id; const int32_t* sample_id;
};

__global__ __launch_bounds__(256) void attn_cell_bwd_kernel(const AttnCellBwdArgs a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.M * a.H) return;
    const int H = a.H, m = i / H, u = i % H;
    float dout = a.dcat[(size_t)m * a.ld_cat + u];
    if (a.dqs)
        for (int s = 0; s < a.nq; ++s) dout += a.dqs[(size_t)s * a.q_stride + i];
    if (a.keep < 1.0f)
        dout = (dout / a.keep) * dropout_keep01(a.seed_lo, a.seed_hi, (uint32_t)a.video_id[m], (uint32_t)a.sample_id[m], a.drop_code,
                                               (uint32_t)u, a.keep);
    float dh = dout;
    if (a.dxs)
        for (int s = 0; s < a.nx; ++s) dh += a.dxs[(size_t)s * a.x_stride + (size_t)m * a.ld_x + a.x_col0 + u];
    const float* g = a.gates + (size_t)m * 4 * H + u;
    const float si = g[0], tj = g[H], sf = g[2 * H], so = g[3 * H];
    const float tc = dm_tanhf(a.c_new[i]);
    const float cp = a.c_prev[i];
    float dc = dh * so * (1.f - tc * tc);
    if (a.dc_in) dc += a.dc_in[i];
    float* z = a.dz + (size_t)m * 4 * H + u;
    z[0] = dc * tj * si * (1.f - si);
    z[H] = dc * si * (1.f - tj * tj);
    z[2 * H] = dc * cp * sf * (1.f - sf);
    z[3 * H] = dh * tc * so * (1.f - so);
    a.dc_out[i] = dc * sf;
}

// loss = (sum coef * nll + sum reg_coef * max(0, m - asum)) / sum(mask)  (original_attention.py:144-149), 1 / the global sum(mask)
// for the gradient bucket, a zeroed ||g||^2 accumulator: one workgroup, deterministic.
__global__ __launch_bounds__(256) void attn_step_scalars_kernel(const float* coef, const float* nll, const float* reg_coef, const float* asum,
                                                                float reg_m, int R, const float* msum_local, const float* gsum_global,
                                                                float* loss, float* gscale, float* sumsq)
{
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < R; i += 256) {
        s += (double)coef[i] * (double)nll[i];
        if (reg_coef) {
            const float hinge = reg_m - asum[i];
            if (hinge > 0.f) s += (double)(reg_coef[i] * hinge);
        }
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (loss) *loss = (float)(sh[0] / (double)*msum_local);
        if (gscale) *gscale = 1.0f / *gsum_global;
        if (sumsq) *sumsq = 0.f;
    }
}

// caption [B, Tc] / mask [B, Tc] (row-major, as fed) -> what the loss kernels take, time-major: target_tm[t*B + b], coef_tm[t*B + b] =
// mask[b, t], reg_tm = beta * mask (optional), *mask_sum = sum(mask): ONE launch of one workgroup (deterministic sum) instead of
// four tensor-library kernels per step.
__global__ __launch_bounds__(256) void attn_loss_inputs_kernel(const int32_t* cap, const float* mask, int B, int Tc, float beta, int32_t* target_tm,
                                                               float* coef_tm, float* reg_tm, float* mask_sum)
{
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < B * Tc; i += 256) {
        const int t = i / B, b = i % B;
        const float m = mask[b * Tc + t];
        target_tm[i] = cap[b * Tc + t];
        coef_tm[i] = m;
        if (reg_tm) reg_tm[i] = beta * m;
        s += (double)m;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *mask_sum = (float)sh[0];
}

// Saved activations + backward scratch, carved from the caller's buffer.  Everything is time-major ([step][row]), so a
// truncated unroll (caption_steps < Tc) is the leading part of the full one's layout.
struct AttnWs {
    int32_t *encidx, *prev, *tgt, *vid, *sid;
    float *Vt, *P;                 // [Tv*B, H] frame embeddings (time-major) and the hoisted image part
    float *hWa, *alpha, *asum, *ctx;   // [Tc][B][H], [Tc][Tv][B], [Tc][B], [Tc][B][H]
    float *G3, *C3, *H3, *O3;      // gates [Tc][B][4H] (first the hoisted partial); states / dropped outputs [(Tc+1)][B][H], slot t+1 = step t
    float* Y;                      // [Tc*B, H] output layer
    float *dY, *dcat, *dZ3, *dxs, *dqs, *dc, *dhWa, *dEmb, *dPt, *dVtt, *dEv;
    float* bslab; size_t bslab_floats;    // split-K slabs of the batched data-gradient products
    unsigned long long* packed;    // greedy picks [Tc][B][kPickStride]
    float* aimg; unsigned* async_; // persistent forward recurrence (attn_chain.hip): fragment images, hand-off counters
    float *bimg, *bex, *brow_; unsigned* bsync;   // persistent backward recurrence (attn_chain_bwd.hip)
    float* deh;                                   // ... its d(score) history [Tc][Tv][B] (more than 5 frames: dP / dV are accumulated behind the launch)
};

size_t carve_attn(Carver& c, const s2vt_dims* d, int B, AttnWs* out)
{
    const size_t H = d->lstm_dim, V = d->n_words, Tv = d->n_video_lstm_step, Tc = d->n_caption_lstm_step, b = B;
    AttnWs w;
    w.encidx = c.take<int32_t>(Tv * b); w.prev = c.take<int32_t>(Tc * b); w.tgt = c.take<int32_t>(Tc * b);
    w.vid = c.take<int32_t>(b); w.sid = c.take<int32_t>(b);
    w.Vt = c.take<float>(Tv * b * H); w.P = c.take<float>(Tv * b * H);
    w.hWa = c.take<float>(Tc * b * H); w.alpha = c.take<float>(Tc * Tv * b); w.asum = c.take<float>(Tc * b); w.ctx = c.take<float>(Tc * b * H);
    w.G3 = c.take<float>(Tc * b * 4 * H); w.C3 = c.take<float>((Tc + 1) * b * H); w.H3 = c.take<float>((Tc + 1) * b * H);
    w.O3 = c.take<float>((Tc + 1) * b * H);
    w.Y = c.take<float>(Tc * b * H);
    w.dY = c.take<float>(Tc * b * H); w.dcat = c.take<float>(Tc * b * 3 * H); w.dZ3 = c.take<float>(Tc * b * 4 * H);
    w.dxs = c.take<float>((size_t)kXSlabs * b * 3 * H); w.dqs = c.take<float>((size_t)kQSlabs * b * H); w.dc = c.take<float>(b * H);
    w.dhWa = c.take<float>(Tc * b * H); w.dEmb = c.take<float>(Tc * b * H);
    w.dPt = c.take<float>(Tv * b * H); w.dVtt = c.take<float>(Tv * b * H); w.dEv = c.take<float>(Tv * b * H);
    w.deh = c.take<float>(Tc * Tv * b);
    {
        // dY = dlogits @ Wout^T ([Tc B, H], K = |V|) and dcat = dpre @ Wp^T ([Tc B, 3H], K = H) when they are short of tiles
        size_t need = 0;
        for (size_t tc = 1; tc <= Tc; ++tc) {
            const int s1 = dx_splits((int)(tc * b), (int)H, (int)V), s2 = dx_splits((int)(tc * b), (int)(3 * H), (int)H);
            const size_t n1 = s1 > 1 ? (size_t)s1 * tc * b * H : 0, n2 = s2 > 1 ? (size_t)s2 * tc * b * 3 * H : 0;
            if (n1 > need) need = n1;
            if (n2 > need) need = n2;
        }
        w.bslab = need ? c.take<float>(need) : nullptr;
        w.bslab_floats = need;
    }
    w.packed = c.take<unsigned long long>(Tc * b * kPickStride);
    w.aimg = c.take<float>(attn_chain_scratch_floats((int)H)); w.async_ = c.take<unsigned>(kAttnChainSyncBytes / 4);
    {
        size_t imgf, exf, rowf, syncb;
        attn_bwd_chain_scratch((int)H, &imgf, &exf, &rowf, &syncb);
        w.bimg = c.take<float>(imgf); w.bex = c.take<float>(exf); w.brow_ = c.take<float>(rowf); w.bsync = c.take<unsigned>(syncb / 4);
    }
    if (out) *out = w;
    return c.off;
}

bool attn_dims_ok(const s2vt_dims* d)
{
    return d && d->dim_image > 0 && d->n_words > 0 && d->lstm_dim > 0 && d->n_video_lstm_step > 0 && d->n_video_lstm_step <= 64 &&
           d->n_caption_lstm_step > 0;
}

bool attn_params_ok(const s2vt_attn_params* p)
{
    return p && p->Wemb && p->encode_image_W && p->encode_image_b && p->embed_att_w && p->embed_att_Wa && p->embed_att_Ua &&
           p->embed_att_ba && p->embed_word_W && p->embed_word_b && p->embed_nn_Wp && p->embed_nn_bp && p->lstm3_W && p->lstm3_b;
}

// frame embedding to H dims in time-major rows (frame t, video b) (:95-98) and the hoisted image part V @ Ua + ba (:107)
int attn_prologue(const s2vt_dims* d, const s2vt_attn_params* p, const float* video, int B, const AttnWs& w, hipStream_t st)
{
    const int H = d->lstm_dim, D = d->dim_image, Tv = d->n_video_lstm_step;
    hipLaunchKernelGGL(attn_enc_index_kernel, dim3((B * Tv + 255) / 256), dim3(256), 0, st, w.encidx, B, Tv);
    HIP_TRY(hipGetLastError());
    ASeg sv = make_seg(video, D, D, 0, 0, w.encidx);
    HIP_TRY(store_call(&sv, 1, p->encode_image_W, H, p->encode_image_b, w.Vt, H, Tv * B, H, 0, -1, st));
    ASeg sp = make_seg(w.Vt, H, H, 0);
    HIP_TRY(store_call(&sp, 1, p->embed_att_Ua, H, p->embed_att_ba, w.P, H, Tv * B, H, 0, -1, st));
    return S2VT_OK;
}

hipError_t attn_step(const s2vt_attn_params* p, const AttnWs& w, int t, int Tv, int B, int H, const float* query, hipStream_t st)
{
    const size_t BH = (size_t)B * H;
    if (t > 0) {       // (step 0: the query is the zero state, h_prev @ Wa = 0, :102)
        ASeg sq = make_seg(query, H, H, 0);
        hipError_t e = store_call(&sq, 1, p->embed_att_Wa, H, nullptr, w.hWa + t * BH, H, B, H, 0, -1, st);
        if (e != hipSuccess) return e;
    }
    AttnFwdArgs a;
    std::memset(&a, 0, sizeof(a));
    a.hWa = t > 0 ? w.hWa + t * BH : nullptr; a.P = w.P; a.Vt = w.Vt; a.w = p->embed_att_w;
    a.alpha = w.alpha + (size_t)t * Tv * B; a.asum = w.asum + (size_t)t * B; a.ctx = w.ctx + t * BH;
    a.Tv = Tv; a.B = B; a.H = H;
    return launch_attn_fwd(a, st);
}

}  // namespace

extern "C" {

size_t s2vt_attn_workspace_bytes(const s2vt_dims* d, int32_t B)
{
    if (!attn_dims_ok(d) || B <= 0) return 0;
    Carver c(nullptr, 0);
    return carve_attn(c, d, B, nullptr);
}

int s2vt_attn_teacher_forced_fwd(const s2vt_dims* d, const s2vt_attn_params* p, const float* video, int32_t B, const int32_t* caption,
                                 int32_t caption_steps, float keep, uint64_t seed, const int32_t* video_id, const int32_t* sample_id,
                                 float* logits, float* alphas_out, void* workspace, size_t workspace_bytes, s2vt_stream stream)
{
    if (!attn_dims_ok(d) || !attn_params_ok(p) || !video || !caption || !logits || !workspace || B <= 0) return S2VT_E_BADARG;
    if (!(keep > 0.0f) || (keep < 1.0f && (!video_id || !sample_id))) return S2VT_E_BADARG;
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) return S2VT_E_ALIGN;
    if (caption_steps < 1 || caption_steps > d->n_caption_lstm_step) return S2VT_E_BADARG;
    if (chain_fault()) return S2VT_E_CHAIN_TIMEOUT;
    const int H = d->lstm_dim, V = d->n_words, Tv = d->n_video_lstm_step, Tc = caption_steps;
    Carver c(workspace, workspace_bytes);
    AttnWs w;
    carve_attn(c, d, B, &w);
    if (!c.ok()) return S2VT_E_WORKSPACE;
    hipStream_t st = S(stream);
    const size_t BH = (size_t)B * H;

    HIP_TRY(launch_prep_caption(caption, w.prev, w.tgt, B, d->n_caption_lstm_step, st));      // prev[t*B + b] = caption[b][t-1] for t >= 1
    {
        ZeroList z;     // zero initial state (:100-101), the zero partial of step 0 (current_embed = 0, :105)
        z.add(w.C3, BH * 4); z.add(w.H3, BH * 4); z.add(w.O3, BH * 4); z.add(w.G3, BH * 4 * 4);
        HIP_TRY(launch_zero_regions(z, st));
    }
    int rc = attn_prologue(d, p, video, B, w, st);
    if (rc != S2VT_OK) return rc;
    // the embedding rows of W3 for every step >= 1 in one product, written where the step's gates will go: the first block of
    // each pre-activation chain (the word fed at step t is caption[:, t-1], :141-142)
    if (Tc > 1) {
        ASeg se = make_seg(p->Wemb, H, H, H, 0, w.prev + B);
        HIP_TRY(store_call(&se, 1, p->lstm3_W, 4 * H, nullptr, w.G3 + 4 * BH, 4 * H, (Tc - 1) * B, 4 * H, 0, -1, st));
    }
    NoiseIds ids{video_id, sample_id, seed};
    if (attn_chain_eligible(B, H, Tv) && !chain_fault() && !(reinterpret_cast<uintptr_t>(p->lstm3_W) & 15) && !(reinterpret_cast<uintptr_t>(p->embed_att_Wa) & 15)) {
        // the whole recurrence -- query projection, score / softmax / context, LSTM3, all Tc steps -- in ONE persistent launch
        AttnChainLaunch a;
        std::memset(&a, 0, sizeof(a));
        a.W3 = p->lstm3_W; a.ldw = 4 * H; a.b3 = p->lstm3_b;
        a.cinit = w.G3; a.cinit_tstride = (size_t)4 * BH; a.ldcinit = 4 * H;
        a.C = w.C3; a.Hh = w.H3; a.Out = w.O3; a.state_tstride = BH; a.gates = w.G3; a.gates_tstride = (size_t)4 * BH;
        a.Wa = p->embed_att_Wa; a.ldwa = H; a.P = w.P; a.Vt = w.Vt; a.w = p->embed_att_w;
        a.hWa = w.hWa; a.hwa_tstride = BH; a.alpha = w.alpha; a.asum = w.asum; a.ctx = w.ctx;
        a.B = B; a.H = H; a.T = Tc; a.Tv = Tv;
        a.keep = keep; a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.drop_code0 = kDropCode3;
        a.video_id = video_id; a.sample_id = sample_id;
        a.img = w.aimg; a.sync = w.async_;
        HIP_TRY(launch_attn_chain(a, st));
    } else {
        for (int t = 0; t < Tc; ++t) {
            HIP_TRY(attn_step(p, w, t, Tv, B, H, w.O3 + t * BH, st));                                     // (:113-128)
            // LSTM3 (:131): the chain continues from the hoisted partial with the recurrent rows, then the context rows
            ASeg s3[2] = {make_seg(w.H3 + t * BH, H, H, 2 * H), make_seg(w.ctx + t * BH, H, H, 0)};
            HIP_TRY(lstm_call(s3, 2, p->lstm3_W, p->lstm3_b, w.C3 + t * BH, 0, w.C3 + (t + 1) * BH, w.H3 + (t + 1) * BH, w.O3 + (t + 1) * BH,
                              w.G3 + (size_t)t * 4 * BH, B, H, keep, ids, kDropCode3 + (uint32_t)t, -1, st, w.G3 + (size_t)t * 4 * BH, 4 * H, 0));
        }
    }
    // output layer for all steps at once (:134): chain blocks [embed ; atten ; output1]; step 0 has no word
    {
        ASeg s0[2] = {make_seg(w.ctx, H, H, H), make_seg(w.O3 + BH, H, H, 0)};
        HIP_TRY(store_call(s0, 2, p->embed_nn_Wp, H, p->embed_nn_bp, w.Y, H, B, H, 1, -1, st));
        if (Tc > 1) {
            ASeg s1[3] = {make_seg(p->Wemb, H, H, 2 * H, 0, w.prev + B), make_seg(w.ctx + BH, H, H, H), make_seg(w.O3 + 2 * BH, H, H, 0)};
            HIP_TRY(store_call(s1, 3, p->embed_nn_Wp, H, p->embed_nn_bp, w.Y + BH, H, (Tc - 1) * B, H, 1, -1, st));
        }
    }
    // vocabulary logits (:143), rows t*B + b
    ASeg so = make_seg(w.Y, H, H, 0);
    HIP_TRY(store_call(&so, 1, p->embed_word_W, V, p->embed_word_b, logits, V, Tc * B, V, 0, -1, st));
    if (alphas_out) {
        CopyList cl;
        if (cl.add(alphas_out, w.alpha, (size_t)Tc * Tv * B * 4)) HIP_TRY(launch_copy_regions(cl, st));
        else HIP_TRY(hipMemcpyAsync(alphas_out, w.alpha, (size_t)Tc * Tv * B * 4, hipMemcpyDeviceToDevice, st));
    }
    return S2VT_OK;
}

int s2vt_attn_loss_inputs(const int32_t* caption, const float* mask, int32_t B, int32_t Tc, float beta, int32_t* target_tm, float* coef_tm,
                          float* reg_tm, float* mask_sum, s2vt_stream stream)
{
    if (!caption || !mask || !target_tm || !coef_tm || !mask_sum || B <= 0 || Tc <= 0) return S2VT_E_BADARG;
    hipLaunchKernelGGL(attn_loss_inputs_kernel, dim3(1), dim3(256), 0, S(stream), caption, mask, B, Tc, beta, target_tm, coef_tm, reg_tm, mask_sum);
    HIP_TRY(hipGetLastError());
    return S2VT_OK;
}

int s2vt_attn_step_scalars(const float* coef, const float* nll, int64_t R, const float* reg_coef, float reg_m, const float* mask_sum_local,
                           const float* mask_sum_global, float* loss, float* gscale, float* sumsq, const s2vt_dims* d, int32_t B,
                           void* workspace, size_t workspace_bytes, s2vt_stream stream)
{
    if (!coef || !nll || R < 0 || !mask_sum_local || !mask_sum_global || !attn_dims_ok(d) || B <= 0 || !workspace) return S2VT_E_BADARG;
    if (R > (int64_t)d->n_caption_lstm_step * B) return S2VT_E_BADARG;
    Carver c(workspace, workspace_bytes);
    AttnWs w;
    carve_attn(c, d, B, &w);
    if (!c.ok()) return S2VT_E_WORKSPACE;
    hipLaunchKernelGGL(attn_step_scalars_kernel, dim3(1), dim3(256), 0, S(stream), coef, nll, reg_coef, w.asum, reg_m, (int)R, mask_sum_local,
                       mask_sum_global, loss, gscale, sumsq);
    HIP_TRY(hipGetLastError());
    return S2VT_OK;
}

int s2vt_attn_bptt_bwd(const s2vt_dims* d, const s2vt_attn_params* p, const s2vt_attn_params* grads, const float* video, int32_t B,
                       const float* dlogits, int32_t caption_steps, const float* reg_coef, float reg_m, float keep, uint64_t seed,
                       const int32_t* video_id, const int32_t* sample_id, void* workspace, size_t workspace_bytes, s2vt_stream stream)
{
    if (!attn_dims_ok(d) || !attn_params_ok(p) || !attn_params_ok(grads) || !video || !dlogits || !workspace || B <= 0) return S2VT_E_BADARG;
    if (!(keep > 0.0f) || (keep < 1.0f && (!video_id || !sample_id))) return S2VT_E_BADARG;
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) return S2VT_E_ALIGN;
    if (caption_steps < 1 || caption_steps > d->n_caption_lstm_step) return S2VT_E_BADARG;
    if (d->lstm_dim & 3) return S2VT_E_BADARG;
    const int H = d->lstm_dim, V = d->n_words, D = d->dim_image, Tv = d->n_video_lstm_step, Tc = caption_steps;
    Carver c(workspace, workspace_bytes);
    AttnWs w;
    carve_attn(c, d, B, &w);
    if (!c.ok()) return S2VT_E_WORKSPACE;
    hipStream_t st = S(stream);
    const size_t BH = (size_t)B * H;
    const int R = Tc * B, R1 = (Tc - 1) * B;

    const bool persistent = attn_bwd_chain_eligible(B, H, Tv) && !chain_fault() && !(reinterpret_cast<uintptr_t>(p->lstm3_W) & 15) &&
                            !(reinterpret_cast<uintptr_t>(p->embed_att_Wa) & 15);
    // Gated overlap (S2VT_OVERLAP=0 switches it off): the weight gradients of the vocabulary projection and of the output layer feed
    // nothing in the recurrence -- with the persistent backward recurrence they are launched on the side stream once its grid is resident and
    // run BESIDE it (a one-wave-per-SIMD grid that waits in hand-offs more than half of its time), joined at the end of the call.
    // Measured (bench.py --workload attention / attention32, S2VT_OVERLAP=0 against 2, twice each): Tv = 5: 2.63 -> 2.57 ms per step; Tv = 32: 3.25 -> 3.27.
    // This recurrence holds 150 KB of LDS per CU, so the LDS-staged contractions cannot share a CU with it (as they do with the LSTM recurrences'
    // 64 KB): they fill the CUs its workgroups leave at the end and run beside the launches that follow it.  On for the register-frames form only.
    SideStream& ss = side_stream();
    const bool gated = ss.ok && ss.mode == 2 && persistent && Tv <= 5;
    std::unique_lock<std::mutex> side_lk(side_stream_mutex(), std::defer_lock);
    if (gated) side_lk.lock();
    ChainGate gate{ss.s, ss.ev[3], false};
    // ---- vocabulary projection: dWout, dbout, d(output layer)
    TnArgs dwout{w.Y, nullptr, H, dlogits, V, grads->embed_word_W, V, R, H, V, 1};
    dwout.colsum = grads->embed_word_b;
    if (!gated) HIP_TRY(launch_gemm_tn(dwout, st));
    HIP_TRY(nn_bwd_slabs(dlogits, V, p->embed_word_W, V, w.dY, H, R, H, V, w.bslab, st, w.bslab_floats));
    hipLaunchKernelGGL(attn_tanh_bwd_kernel, dim3((unsigned)(((size_t)R * H / 4 + 255) / 256)), dim3(256), 0, st, w.dY, w.Y, (size_t)R * H / 4);
    HIP_TRY(hipGetLastError());
    // ---- output layer: Wp rows [output1 ; atten ; current_embed], its bias, and d[out | ctx | emb] for every step at once
    auto output_layer_grads = [&](hipStream_t s) -> int {
        TnArgs a{w.O3 + BH, nullptr, H, w.dY, H, grads->embed_nn_Wp, H, R, H, H, 1};
        a.colsum = grads->embed_nn_bp;
        HIP_TRY(launch_gemm_tn(a, s));
        TnArgs b{w.ctx, nullptr, H, w.dY, H, grads->embed_nn_Wp + (size_t)H * H, H, R, H, H, 1};
        HIP_TRY(launch_gemm_tn(b, s));
        if (Tc > 1) {
            TnArgs e{p->Wemb, w.prev + B, H, w.dY + BH, H, grads->embed_nn_Wp + (size_t)2 * H * H, H, R1, H, H, 1};
            e.gather_rows = V;
            HIP_TRY(launch_gemm_tn(e, s));
        }
        return S2VT_OK;
    };
    if (!gated) { const int rc = output_layer_grads(st); if (rc != S2VT_OK) return rc; }
    HIP_TRY(nn_bwd_slabs(w.dY, H, p->embed_nn_Wp, H, w.dcat, 3 * H, R, 3 * H, H, w.bslab, st, w.bslab_floats));
    {
        ZeroList z;
        z.add(w.dPt, (size_t)Tv * BH * 4); z.add(w.dVtt, (size_t)Tv * BH * 4);
        HIP_TRY(launch_zero_regions(z, st));
    }
    // ---- the recurrence, back through time
    if (persistent) {
        // ONE persistent launch: cell backward, dz @ [W3 h rows ; W3 context rows]^T, attention backward, dhWa @ Wa^T, all Tc steps
        AttnBwdChainLaunch a;
        std::memset(&a, 0, sizeof(a));
        a.W3 = p->lstm3_W; a.ldw = 4 * H; a.Wa = p->embed_att_Wa; a.ldwa = H;
        a.gates = w.G3; a.gates_tstride = (size_t)4 * BH; a.C = w.C3; a.state_tstride = BH;
        a.dcat = w.dcat; a.dcat_tstride = (size_t)3 * BH; a.ld_cat = 3 * H; a.dZ = w.dZ3; a.dz_tstride = (size_t)4 * BH;
        a.hWa = w.hWa; a.hwa_tstride = BH; a.P = w.P; a.Vt = w.Vt; a.w = p->embed_att_w; a.alpha = w.alpha;
        a.reg_coef = reg_coef; a.asum = w.asum; a.reg_m = reg_m;
        a.dhWa = w.dhWa; a.dhwa_tstride = BH; a.dP = w.dPt; a.dVt = w.dVtt; a.dw = grads->embed_att_w;
        a.deh = w.deh;
        a.B = B; a.H = H; a.T = Tc; a.Tv = Tv;
        a.keep = keep; a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.drop_code0 = kDropCode3;
        a.video_id = video_id; a.sample_id = sample_id;
        a.img = w.bimg; a.ex = w.bex; a.dctxs = w.brow_; a.sync = w.bsync;
        if (gated) chain_gate_arm(&gate);
        const hipError_t re = launch_attn_bwd_chain(a, st);
        chain_gate_arm(nullptr);
        HIP_TRY(re);
        if (gated) {
            // (reads: dlogits, Y, dY, O3, ctx, Wemb -- final since before the launch; writes: the gradients of embed_word_W/b and embed_nn_Wp/bp,
            //  which nothing else in this call touches.  A gate that did not fire -- a zero-step launch -- puts the side stream behind the caller's)
            if (!gate.fired) HIP_TRY(fork_to(st, ss.s, ss.ev[3]));
            HIP_TRY(launch_gemm_tn(dwout, ss.s));
            const int rc = output_layer_grads(ss.s);
            if (rc != S2VT_OK) return rc;
        }
        // the embedding block of dz @ W3^T does not feed the recurrence: all steps >= 1 at once, on top of the output layer's block
        if (Tc > 1) {
            ASeg sz = make_seg(w.dZ3 + 4 * BH, 4 * H, 4 * H, 0);
            HIP_TRY(store_call(&sz, 1, p->lstm3_W + (size_t)H * 4 * H, 4 * H, nullptr, w.dEmb + BH, H, R1, H, 0, -1, st, w.dcat + 3 * BH + 2 * H, 3 * H, true));
        }
    } else {
        // split-K plans of the two per-step data-gradient products (order-free): enough slabs for >= ~512 workgroups
        int sx = (512 + ((3 * H + 63) / 64) - 1) / ((3 * H + 63) / 64) / ((B + 63) / 64);
        if (sx < 1) sx = 1;
        if (sx > kXSlabs) sx = kXSlabs;
        const int kperx = ((4 * H + sx - 1) / sx + BK - 1) / BK * BK, nx = (4 * H + kperx - 1) / kperx;
        int sq = kQSlabs;
        while (sq > 1 && H / sq < 128) --sq;
        const int kperq = ((H + sq - 1) / sq + BK - 1) / BK * BK, nq = (H + kperq - 1) / kperq;
        for (int t = Tc - 1; t >= 0; --t) {
            const bool last = t == Tc - 1;
            AttnCellBwdArgs a;
            std::memset(&a, 0, sizeof(a));
            a.gates = w.G3 + (size_t)t * 4 * BH; a.c_new = w.C3 + (t + 1) * BH; a.c_prev = w.C3 + t * BH;
            a.dcat = w.dcat + (size_t)t * 3 * BH; a.ld_cat = 3 * H;
            a.dqs = last ? nullptr : w.dqs; a.nq = nq; a.q_stride = BH;
            a.dxs = last ? nullptr : w.dxs; a.nx = nx; a.x_stride = 3 * BH; a.ld_x = 3 * H; a.x_col0 = 2 * H;
            a.dc_in = last ? nullptr : w.dc; a.dc_out = w.dc; a.dz = w.dZ3 + (size_t)t * 4 * BH;
            a.M = B; a.H = H; a.keep = keep; a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.drop_code = kDropCode3 + (uint32_t)t;
            a.video_id = video_id; a.sample_id = sample_id;
            hipLaunchKernelGGL(attn_cell_bwd_kernel, dim3((B * H + 255) / 256), dim3(256), 0, st, a);
            HIP_TRY(hipGetLastError());
            // d[ctx | emb | h_prev] = dz @ W3^T as split-K slabs: the attention backward sums the ctx and emb blocks, the next
            // (earlier) step's cell backward the h block
            HIP_TRY(nn_bwd(w.dZ3 + (size_t)t * 4 * BH, 4 * H, p->lstm3_W, 4 * H, w.dxs, 3 * H, B, 3 * H, 4 * H, sx, 3 * BH, st, sx > 1 ? kSlabTileCfg : -1));
            AttnBwdArgs g;
            std::memset(&g, 0, sizeof(g));
            g.hWa = t > 0 ? w.hWa + t * BH : nullptr; g.P = w.P; g.Vt = w.Vt; g.w = p->embed_att_w; g.alpha = w.alpha + (size_t)t * Tv * B;
            g.dctx = w.dcat + (size_t)t * 3 * BH + H; g.ld_dctx = 3 * H;
            g.slabs = w.dxs; g.nslab = nx; g.slab_stride = 3 * BH; g.ld_slab = 3 * H; g.ctx_col0 = 0; g.emb_col0 = H;
            if (t > 0) { g.demb_dense = w.dcat + (size_t)t * 3 * BH + 2 * H; g.ld_demb = 3 * H; g.demb_out = w.dEmb + t * BH; }
            if (reg_coef) { g.reg_coef = reg_coef + (size_t)t * B; g.asum = w.asum + (size_t)t * B; g.reg_m = reg_m; }
            g.dhWa = t > 0 ? w.dhWa + t * BH : nullptr; g.dP = w.dPt; g.dVt = w.dVtt; g.acc = 1; g.dw = grads->embed_att_w;
            g.Tv = Tv; g.B = B; g.H = H;
            HIP_TRY(launch_attn_bwd(g, st));
            // gradient w.r.t. the previous step's dropped output through this step's query: dhWa @ Wa^T (slabs, summed by the cell backward)
            if (t > 0) HIP_TRY(nn_bwd(w.dhWa + t * BH, H, p->embed_att_Wa, H, w.dqs, H, B, H, H, sq, BH, st, sq > 1 ? kSlabTileCfg : -1));
        }
    }
    // ---- weight gradients of the recurrence, one contraction over all unrolled steps per block
    {
        TnArgs a{w.ctx, nullptr, H, w.dZ3, 4 * H, grads->lstm3_W, 4 * H, R, H, 4 * H, 1};                               // rows [0, H): atten
        a.colsum = grads->lstm3_b;
        HIP_TRY(launch_gemm_tn(a, st));
        TnArgs h{w.H3, nullptr, H, w.dZ3, 4 * H, grads->lstm3_W + (size_t)2 * H * 4 * H, 4 * H, R, H, 4 * H, 1};        // rows [2H, 3H): h_prev
        HIP_TRY(launch_gemm_tn(h, st));
        if (Tc > 1) {
            TnArgs e{p->Wemb, w.prev + B, H, w.dZ3 + 4 * BH, 4 * H, grads->lstm3_W + (size_t)H * 4 * H, 4 * H, R1, H, 4 * H, 1};   // rows [H, 2H): current_embed
            e.gather_rows = V;
            HIP_TRY(launch_gemm_tn(e, st));
            TnArgs q{w.O3 + BH, nullptr, H, w.dhWa + BH, H, grads->embed_att_Wa, H, R1, H, H, 1};                       // query of step t = out of step t-1
            HIP_TRY(launch_gemm_tn(q, st));
            HIP_TRY(launch_scatter_add_rows(w.dEmb + BH, H, w.prev + B, R1, H, grads->Wemb, H, st));                    // tf.nn.embedding_lookup (:141-142)
        }
    }
    // ---- image part P = V @ Ua + ba and the frame embedding V = video @ encode_image_W + b
    {
        TnArgs u{w.Vt, nullptr, H, w.dPt, H, grads->embed_att_Ua, H, Tv * B, H, H, 1};
        u.colsum = grads->embed_att_ba;
        HIP_TRY(launch_gemm_tn(u, st));
        ASeg sp = make_seg(w.dPt, H, H, 0);
        HIP_TRY(store_call(&sp, 1, p->embed_att_Ua, H, nullptr, w.dEv, H, Tv * B, H, 0, -1, st, w.dVtt, H, true));     // dV = dV(ctx path) + dP @ Ua^T
        TnArgs v{video, w.encidx, D, w.dEv, H, grads->encode_image_W, H, Tv * B, D, H, 1};
        v.gather_rows = Tv * B;
        v.colsum = grads->encode_image_b;
        HIP_TRY(launch_gemm_tn(v, st));
    }
    if (gated) HIP_TRY(fork_to(ss.s, st, ss.ev[2]));        // join: the caller's stream waits for the side stream's gradients
    return S2VT_OK;
}

int s2vt_attn_decode_greedy(const s2vt_dims* d, const s2vt_attn_params* p, const float* video, int32_t B, int32_t video_base, int32_t* ids_out,
                            float* alphas_out, void* workspace, size_t workspace_bytes, s2vt_stream stream)
{
    if (!attn_dims_ok(d) || !attn_params_ok(p) || !video || !ids_out || !workspace || B <= 0) return S2VT_E_BADARG;
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) return S2VT_E_ALIGN;
    const int H = d->lstm_dim, V = d->n_words, Tv = d->n_video_lstm_step, Tc = d->n_caption_lstm_step;
    Carver c(workspace, workspace_bytes);
    AttnWs w;
    carve_attn(c, d, B, &w);
    if (!c.ok()) return S2VT_E_WORKSPACE;
    hipStream_t st = S(stream);
    const size_t BH = (size_t)B * H;
    {
        ZeroList z;
        z.add(w.C3, BH * 4); z.add(w.H3, BH * 4); z.add(w.packed, (size_t)Tc * B * kPickStride * 8);
        HIP_TRY(launch_zero_regions(z, st));
    }
    hipLaunchKernelGGL(attn_rows_kernel, dim3((B + 255) / 256), dim3(256), 0, st, w.vid, w.sid, B, video_base);
    HIP_TRY(hipGetLastError());
    int rc = attn_prologue(d, p, video, B, w, st);
    if (rc != S2VT_OK) return rc;
    NoiseIds none{nullptr, nullptr, 0};
    NoiseIds ids{w.vid, w.sid, 0};
    for (int t = 0; t < Tc; ++t) {
        // no dropout in the samplers (self.lstm3, not lstm3_dropout, :188,:235): the query is the clean h
        HIP_TRY(attn_step(p, w, t, Tv, B, H, w.H3 + t * BH, st));
        const unsigned long long* tok = t > 0 ? w.packed + (size_t)(t - 1) * B * kPickStride : nullptr;    // the word picked at step t-1 (:196-197)
        if (t == 0) {
            ASeg s3 = make_seg(w.ctx, H, H, 0);
            HIP_TRY(lstm_call(&s3, 1, p->lstm3_W, p->lstm3_b, w.C3, 0, w.C3 + BH, w.H3 + BH, nullptr, nullptr, B, H, 1.0f, none, 0, -1, st));
            ASeg sy[2] = {make_seg(w.ctx, H, H, H), make_seg(w.H3 + BH, H, H, 0)};
            HIP_TRY(store_call(sy, 2, p->embed_nn_Wp, H, p->embed_nn_bp, w.Y, H, B, H, 1, -1, st));
        } else {
            ASeg s3[3] = {make_seg(p->Wemb, H, H, H, 0, nullptr, tok, kPickStride), make_seg(w.H3 + t * BH, H, H, 2 * H), make_seg(w.ctx + t * BH, H, H, 0)};
            HIP_TRY(lstm_call(s3, 3, p->lstm3_W, p->lstm3_b, w.C3 + t * BH, 0, w.C3 + (t + 1) * BH, w.H3 + (t + 1) * BH, nullptr, nullptr, B, H, 1.0f,
                              none, 0, -1, st));
            ASeg sy[3] = {make_seg(p->Wemb, H, H, 2 * H, 0, nullptr, tok, kPickStride), make_seg(w.ctx + t * BH, H, H, H),
                          make_seg(w.H3 + (t + 1) * BH, H, H, 0)};
            HIP_TRY(store_call(sy, 3, p->embed_nn_Wp, H, p->embed_nn_bp, w.Y + t * BH, H, B, H, 1, -1, st));
        }
        HIP_TRY(pick_call(w.Y + t * BH, H, p->embed_word_W, p->embed_word_b, B, H, V, ids, t, w.packed + (size_t)t * B * kPickStride, nullptr, -1, st,
                          kPickStride));
    }
    hipLaunchKernelGGL(attn_unpack_ids_kernel, dim3((B * Tc + 255) / 256), dim3(256), 0, st, w.packed, ids_out, B, Tc, kPickStride);
    HIP_TRY(hipGetLastError());
    if (alphas_out) {
        CopyList cl;
        if (cl.add(alphas_out, w.alpha, (size_t)Tc * Tv * B * 4)) HIP_TRY(launch_copy_regions(cl, st));
        else HIP_TRY(hipMemcpyAsync(alphas_out, w.alpha, (size_t)Tc * Tv * B * 4, hipMemcpyDeviceToDevice, st));
    }
    return S2VT_OK;
}

}  // extern "C"
