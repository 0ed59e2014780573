// internal.h -- launcher interfaces shared by the translation units of libs2vt_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_mfma.h"

namespace s2vt {

// forward contraction (exact ascending-k chain); epi in {EPI_STORE, EPI_LSTM, EPI_PICK}.
// cfg < 0: pick a tile configuration from the shape; >= 0 forces table entry `cfg` (tests, tuning).
hipError_t launch_gemm(const GemmArgs& a, int epi, int cfg, hipStream_t st);

// ---- in-library launch profiler (bench.py's roofline leg): when enabled, every contraction launch
// is bracketed by two hipEvents ON ITS OWN STREAM and tallied per (class, tile cfg) with its flops.
// class: 0 STORE, 1 LSTM, 2 PICK, 3 TN.  Collect only after the stream has been synchronised.
struct ProfRow { int cls, cfg; long launches; double ms, flops; const char* name; };
void prof_enable(bool on);
bool prof_on();
bool prof_wants(int cls, int cfg);          // profiler on and (cls, cfg) passes the filter
void prof_filter(int cls, int cfg);         // -1 = any
void prof_record(int cls, int cfg, const char* name, double flops, hipEvent_t e0, hipEvent_t e1);
hipError_t prof_events(hipEvent_t* e0, hipEvent_t* e1);
int prof_collect(ProfRow* rows, int max_rows);
int gemm_num_cfgs(int epi);
const char* gemm_cfg_name(int epi, int cfg);

// C[Kout, N] (+)= sum_m A[row(m), k] * B[m, n]   (weight gradients; order-free, fp32 MFMA)
struct TnArgs {
    const float* A; const int* rowidx; int lda;       // [Mred, Kout] (rows optionally gathered)
    const float* B; int ldb;                           // [Mred, N]
    float* C; int ldc;                                 // [Kout, N]
    int Mred, Kout, N;
    int accumulate;                                    // 1: C += result (fp32 atomics when split)
    float* colsum;                                     // optional [N]: += column sums of B (bias gradient), see launch_gemm_tn
    int gather_rows;                                   // rows of the table `rowidx` indexes (0 = not stated by the caller): the vector paths form
                                                       // rowidx * lda * 4 in 32 bits, so a table known to reach 2 GiB takes the scalar (64-bit) path
};
// colsum: when given, the column sums of B are ADDED to it -- inside the contraction on the vector path, by a colsum launch
// otherwise (so callers never launch one themselves).
hipError_t launch_gemm_tn(const TnArgs& a, hipStream_t st);

hipError_t launch_prep_caption(const int32_t* cap, int32_t* prev, int32_t* tgt, int N, int Tc, hipStream_t st);
hipError_t launch_caption_mask(const int32_t* ids, int N, int Tc, float* mask, int32_t* target_tm, float* mask_sum, float* mask_sum_copy, hipStream_t st);
hipError_t launch_xe_prep(const float* mask, const int32_t* cap, int N, int Tc, float loss_weight, float n_glob, int q1, float* coef_tm,
                          int32_t* target_tm, float* msum, hipStream_t st);
hipError_t launch_mixed_prep(const float* mask, const float* gmask, const float* rewards, const float* baseline, const int32_t* cap,
                             const int32_t* gcap, int Ns, int B, int Tc, float one_minus_lam, float lam, float loss_weight, int q1,
                             float smoothing, float n_glob_b, float* coef_tm, float* smooth_tm, int32_t* cap_all, int32_t* target_tm, float* sums, hipStream_t st);
hipError_t launch_mixed_loss(const float* coef, const float* nll, const int32_t* live_rows, int R, int N, int Ns, float* out3, hipStream_t st);
hipError_t launch_pg_coef(const float* mask, const float* rewards, const float* baseline, float scale, int N, int Tc, float* coef_tm,
                          hipStream_t st);
hipError_t launch_step_scalars(const float* coef, const float* nll, int64_t R, const float* msum_local, const float* gsum_global, float* loss,
                               float* gscale, float* sumsq, hipStream_t st);
hipError_t launch_softmax_nll(float* logits, int ld, int R, int V, const int32_t* target, const float* coef,
                              float smoothing, float* nll, float* lp_t, hipStream_t st, const float* smooth_rows = nullptr);
hipError_t launch_softmax_unshifted_argmax(const float* logits, int ld, int R, int V, int32_t* ids, float* probs, hipStream_t st);
hipError_t launch_lstm_bwd_pointwise(const float* gates, const float* c_new, const float* c_prev, const float* dh_rec,
                                     int nslab, size_t slab_stride, const float* dout_ext, int ld_ext, const float* dc_in,
                                     float* dc_out, float* dz, int M, int H, float keep, uint64_t seed, uint32_t drop_code,
                                     const int32_t* video_id, const int32_t* sample_id, hipStream_t st);
// out[t][n][u] = keep < 1 ? (h[t][n % B][u] / keep) * mask(n, t, u) : h[...]   (DropoutWrapper output of a cell whose
// state trajectory is shared by the rep = N / B sample rows of a video); code = code_base + t
hipError_t launch_expand_dropout(const float* h, float* out, int T, int B, int N, int H, float keep, uint64_t seed,
                                 uint32_t code_base, const int32_t* video_id, const int32_t* sample_id, hipStream_t st);
// dh[t][j][u] = sum_k (dout[t][k*B + j][u] / keep) * mask(k*B + j, t, u)   (adjoint of the above; dout rows ld_out apart)
hipError_t launch_reduce_dropout(const float* dout, int ld_out, float* dh, int T, int B, int N, int H, float keep,
                                 uint64_t seed, uint32_t code_base, const int32_t* video_id, const int32_t* sample_id,
                                 hipStream_t st);
hipError_t launch_colsum(const float* X, int ld, int M, int N, float* out, hipStream_t st);
hipError_t launch_sum_slabs(float* dst, const float* slabs, int nslab, size_t stride, size_t n, hipStream_t st);
// up to 8 buffers zeroed by one launch (sizes in 4-byte words)
struct ZeroList {
    uint32_t* p[8]; size_t n[8]; int count;
    ZeroList() : count(0) {}
    void add(void* ptr, size_t bytes) { if (ptr && bytes) { p[count] = static_cast<uint32_t*>(ptr); n[count] = bytes / 4; ++count; } }
};
hipError_t launch_zero_regions(const ZeroList& z, hipStream_t st);
// up to 4 device-to-device copies by one launch (16-byte aligned pointers, byte counts multiples of 16; anything else: false)
struct CopyList {
    uint4* dst[4]; const uint4* src[4]; size_t n16[4]; int count;
    CopyList() : count(0) {}
    bool add(void* d, const void* s, size_t bytes)
    {
        if (!bytes) return true;
        if (count >= 4 || ((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(s) | bytes) & 15u)) return false;
        dst[count] = static_cast<uint4*>(d); src[count] = static_cast<const uint4*>(s); n16[count] = bytes / 16; ++count;
        return true;
    }
};
hipError_t launch_copy_regions(const CopyList& c, hipStream_t st);
hipError_t launch_gather_rows(const float* src, int ld, const int32_t* idx, int R, int C, float* dst, int ldd, hipStream_t st);   // dst[r,:] = src[idx[r],:]
hipError_t launch_scatter_rows(const float* src, int ld, const int32_t* idx, int R, int C, float* dst, int ldd, hipStream_t st);  // dst[idx[r],:] = src[r,:]
hipError_t launch_gather_i32(const int32_t* src, const int32_t* idx, int R, int32_t* dst, hipStream_t st);
hipError_t launch_scatter_add_rows(const float* dE, int ld, const int32_t* idx, int R, int E, float* dW, int ldw,
                                   hipStream_t st);
hipError_t launch_transpose(const float* in, int ldi, float* out, int ldo, int R, int Cc, hipStream_t st);
hipError_t launch_grad_finalize(float* g, const float* theta, int64_t n, const float* gscale, float wd, float* sumsq,
                                hipStream_t st);
// fault: optional device word -- nonzero makes the launch a no-op (a persistent recurrence timed out upstream: the
// gradients are garbage); applied_step: optional device word that receives `step` when the update IS applied
hipError_t launch_adam_tf(float* theta, const float* g, float* m, float* v, int64_t n, const float* sumsq, float clip,
                          float lr_t, float b1, float b2, float eps, hipStream_t st, const unsigned* fault = nullptr,
                          int32_t* applied_step = nullptr, int32_t step = 0);

// ---- a whole LSTM recurrence (T steps, M <= 64 rows) in one persistent launch (chain.hip)
struct ChainArgs {
    const float* W; int ldw; int kw0;                 // full cell matrix [*, 4H] (ldw floats per row); the recurrent rows start at kw0
    const float* bias;                                 // [4H]
    const float* cinit; size_t cinit_tstride; int ldcinit; int cinit_steps;   // carried partial of step t < cinit_steps: cinit + t*tstride, rows ldcinit apart; NULL = none
    const float* h0; const float* c0;                  // initial state [M, H] or NULL = zeros
    float* C; float* Hh; size_t state_tstride;         // state histories: step t writes slot t + 1 (C + (t+1)*state_tstride)
    float* gates; size_t gates_tstride;                // optional activated gates [T][M][4H] (si | tj | sf | so)
    float* out; size_t out_tstride;                    // optional DropoutWrapper output [T][M][H]
    int M, H, T;
    float keep; uint32_t seed_lo, seed_hi, drop_code0; // dropout of `out`: code = drop_code0 + t
    const int32_t* video_id; const int32_t* sample_id;
    float* abuf;                                       // chain_scratch_floats(H) floats, 16-byte aligned
    unsigned* sync;                                    // kChainSyncBytes bytes: arrival counters + timeout word
    unsigned* status;                                  // (set by the launcher) host-mapped count of timed-out waits
    unsigned* fault;                                   // (set by the launcher) device-resident fault word: 1 after a timeout until chain_ack()
    unsigned spin_limit;                               // (set by the launcher) polls before a grid-wide wait gives up
    int ncg, tpp, img_tiles;                           // (set by the launcher) column groups, row tiles per row part, row tiles of one state image
    // optional (the register-weights form, 257-384 rows; ignored -- a dense launch, same results -- elsewhere): rows sorted by length.
    // perm[v] = row of every per-row array that virtual row v stands for (a permutation of 0 .. M-1), nlive[t] = virtual rows
    // 0 .. nlive[t]-1 are computed at step t (non-increasing in t); the other rows' histories are NOT written at that step
    const int32_t* perm; const int32_t* nlive;
};
bool chain_live_capable(int M, int H);                 // the forward AND backward recurrences of this shape take perm / nlive
hipError_t launch_row_order(const int32_t* live_rows, int n_live, int N, int Tv, int Tc, int32_t* perm, int32_t* nlive, hipStream_t st);
constexpr size_t kChainSyncBytes = 9 * 128;            // 8 counter shards + the status line, a block of its own (multiple of 16)
bool chain_eligible(int M, int H);                     // shape fits the persistent form on this device (and S2VT_CHAIN != 0)
size_t chain_scratch_floats(int H);
hipError_t launch_lstm_chain(const ChainArgs& a, hipStream_t st);
unsigned chain_timeouts();                             // grid-wide waits that gave up, all launches of this process (0 = healthy)
bool chain_operands_ok(const float* W, int ldw, const float* abuf);   // alignment the persistent form needs (else: per-step launches)
bool chain_fault();                                    // a timeout has happened and has not been acknowledged: results since are suspect
const unsigned* chain_fault_word();                    // device word of the current device (1 = fault), NULL when the persistent form is unavailable
hipError_t chain_ack(bool disable);                    // acknowledge (device idle!); disable: per-step launches for the rest of the process

// ---- gated overlap (train.hip, S2VT_OVERLAP=2): a side stream that is released once a persistent grid is RESIDENT.  A persistent recurrence at
// <= 256 rows holds one wave per SIMD at <= 380 VGPRs and spends half its time in hand-offs: an independent weight-gradient contraction can
// share the CUs with it -- provided the recurrence's workgroups got their slots first (a contraction that fills the chip first leaves no
// room for them: the grid would be partly resident and spin until the contraction drains).  The next persistent launch of the arming thread
// records g->ev behind its counter-zeroing kernel and, behind the grid's launch, queues on g->side {wait for g->ev; a one-wave kernel that
// polls the grid's first arrival until every workgroup has arrived, bounded}.  What the caller then launches on g->side starts beside the grid.
struct ChainGate { hipStream_t side; hipEvent_t ev; bool fired; };
void chain_gate_arm(ChainGate* g);                   // nullptr disarms; consumed (fired = true) by the next persistent launch of this thread
hipError_t chain_gate_zeroed(hipStream_t st);        // (launchers) behind the zeroing kernel
hipError_t chain_gate_launched(const unsigned* sync, unsigned arrivals);   // (launchers) behind the grid's launch; arrivals = workgroups arriving at the first hand-off

// ---- the backward recurrence of one cell in one persistent launch (chain.hip): dZ[t] for t = T-1 .. 0
struct BwdChainLaunch {
    const float* W; int ldw; int kw0;                  // cell matrix [*, 4H]; the recurrent rows start at kw0
    const float* gates; size_t gates_tstride;          // activated gates [T][M][4H]
    const float* C; size_t state_tstride;              // cell states [T+1][M][H]
    const float* dext; size_t dext_tstride; int ld_ext; int dext_t0;   // upstream gradient w.r.t. the (dropped) output of step t >= dext_t0; NULL = none
    float* dZ; size_t dz_tstride;                      // [T][M][4H]
    int M, H, T;
    float keep; uint32_t seed_lo, seed_hi, drop_code0;
    const int32_t* video_id; const int32_t* sample_id;
    float* img; float* ex; unsigned* sync;             // scratch, sizes from bwd_chain_scratch (16-byte aligned)
    const int32_t* perm; const int32_t* nlive;         // optional, as ChainArgs: step t runs virtual rows 0 .. nlive[t]-1 only (their dZ rows are the only ones written)
};
bool bwd_chain_eligible(int M, int H);               // the shape fits the persistent form on this device
bool bwd_chain_auto(int M, int H);                   // ... and it is the faster form there (chosen when the caller does not say)
void bwd_chain_scratch(int H, int M, size_t* img_floats, size_t* ex_floats, size_t* sync_bytes);
hipError_t launch_lstm_bwd_chain(const BwdChainLaunch& a, hipStream_t st);

// ---- the sampler's LSTM2 step at 257-384 rows on fragment-order operands (decode4.hip)
struct Dec4Geom { int eg, hg, ech, hch, erow, hgp, ngt, ncg, tpp, img_tiles; };
struct Dec4Launch {
    const float* wemb_p; const float* w2_p; const float* bias;
    const float* cinit; int ldcinit; int cinit_rowmod;
    const unsigned long long* tok; int tok_stride; int tok_const;
    const float* himg_in; float* himg_out;
    const float* c_prev; int cprev_rowmod;
    float* c_new; float* h_new;
    int R, H, E, V;
};
bool decode4_eligible(int R, int H, int E);
void decode4_geometry(int R, int H, int E, Dec4Geom* out);
hipError_t decode4_pack(const float* Wemb, const float* W2, int V, int H, int E, const Dec4Geom& q, float* wemb_p, float* w2_p, hipStream_t st);
hipError_t decode4_state_to_image(const float* h, int B, int R, int H, const Dec4Geom& q, float* img, hipStream_t st);
hipError_t launch_decode_lstm4(const Dec4Launch& a, const Dec4Geom& q, hipStream_t st);

// ---- the sampler's whole decode loop in one persistent launch (decode_loop.hip), on decode4's packed operands
struct DecLoopLaunch {
    const float* wemb_p; const float* w2_p; const float* bias2;
    const float* P2; size_t p2_tstride; int ldp2; int B;      // carried partial of decode step t (row % B)
    const float* c0;                                          // [B, H] cell state at the start of the decoding stage
    float* himg0; float* himg1;                               // state images (himg0 = h at the start, himg1 zeroed)
    unsigned long long* packed; int pick_stride;              // [Tc][R][pick_stride], zeroed
    const float* Wout; int ldwo; const float* bout;
    uint64_t seed; int video_base;                            // noise ids as sampler_rows_kernel: video = video_base + row % B, sample = row / B
    int noise_rows;                                           // rows [0, noise_rows) are sampled (sample id >= 0), the rest argmax
    int R, H, E, V, Tc;
    unsigned* sync;                                           // kChainSyncBytes
};
bool decode_loop_eligible(int R, int H, int E, int V);
hipError_t launch_decode_loop(const DecLoopLaunch& a, const Dec4Geom& q, hipStream_t st);

// ---- temporal attention step (attn.hip): score -> softmax -> context, one workgroup per batch row
struct AttnFwdArgs {
    const float* hWa;            // [B, H] query projection h_prev @ Wa, or NULL = zeros (the first decode step)
    const float* P;              // [Tv, B, H] hoisted image part V @ Ua + ba (original_attention.py:107)
    const float* Vt;             // [Tv, B, H] frame embeddings, time-major (:98)
    const float* w;              // [H]
    float* scores;               // optional [Tv, B]
    float* alpha;                // [Tv, B]
    float* asum;                 // optional [B]: alpha[0] + ... + alpha[min(8, Tv) - 1] (the regulariser's first-8-frames sum, :118,123)
    float* ctx;                  // [B, H]
    int Tv, B, H;
    int RC, ldT, vec;            // (set by the launcher) frames per LDS chunk, LDS row stride, 16-byte path
};
hipError_t launch_attn_fwd(const AttnFwdArgs& a, hipStream_t st);
struct AttnBwdArgs {
    const float* hWa; const float* P; const float* Vt; const float* w; const float* alpha;    // as the forward (hWa may be NULL = zeros)
    const float* dctx; int ld_dctx;                    // dense part of d(ctx) [B, ld] or NULL
    const float* slabs; int nslab; size_t slab_stride; int ld_slab;   // + sum_s slabs[s][b * ld_slab + col0 + h] (split-K slabs of a [B, ld_slab] product)
    int ctx_col0, emb_col0;                            // column of the ctx / embedding block inside a slab row
    const float* demb_dense; int ld_demb; float* demb_out;   // optional: demb_out[b, h] = demb_dense[b, h] + sum_s slabs[..emb_col0 + h]
    const float* reg_coef;                             // optional [B]: beta * mask[b, t] of the alpha regulariser (:123), with ...
    const float* asum; float reg_m;                    // ... the forward's first-8-frames sums [B] and the margin m
    float* dhWa;                                       // [B, H] (optional)
    float* dP; float* dVt; int acc;                    // [Tv, B, H]; acc != 0: += (accumulated over the decode steps)
    float* dw;                                         // [H], atomicAdd
    int Tv, B, H;
    int vec;                                           // (set by the launcher) 16-byte path
};
hipError_t launch_attn_bwd(const AttnBwdArgs& a, hipStream_t st);

// ---- the attention captioner's whole forward recurrence in one persistent launch (attn_chain.hip)
struct AttnChainLaunch {
    const float* W3; int ldw; const float* b3;          // LSTM3 [3H, 4H] (rows [0,H) context, [H,2H) embedding -- hoisted by the caller --, [2H,3H) h)
    const float* cinit; size_t cinit_tstride; int ldcinit;   // the hoisted embedding partial of step t (NULL = none)
    float* C; float* Hh; float* Out; size_t state_tstride;   // histories [T+1][B][H]: step t writes slot t + 1 (slot 0 = the zero state, not touched)
    float* gates; size_t gates_tstride;                 // [T][B][4H] (may alias cinit)
    const float* Wa; int ldwa; const float* P; const float* Vt; const float* w;
    float* hWa; size_t hwa_tstride;                     // [T][B][H], slots 1 .. T-1 written
    float* alpha; float* asum; float* ctx;              // [T][Tv][B], [T][B], [T][B][H]
    int B, H, T, Tv;
    float keep; uint32_t seed_lo, seed_hi, drop_code0; const int32_t* video_id; const int32_t* sample_id;
    float* img;                                         // attn_chain_scratch_floats(H) floats, 16-byte aligned
    unsigned* sync;                                     // kAttnChainSyncBytes
};
constexpr size_t kAttnChainSyncBytes = 3 * kChainSyncBytes;
bool attn_chain_eligible(int B, int H, int Tv);       // the persistent form serves this shape on this device (S2VT_ACHAIN != 0, no hold / fault)
size_t attn_chain_scratch_floats(int H);
hipError_t launch_attn_chain(const AttnChainLaunch& a, hipStream_t st);

// ---- ... and its backward recurrence (attn_chain_bwd.hip)
struct AttnBwdChainLaunch {
    const float* W3; int ldw; const float* Wa; int ldwa;
    const float* gates; size_t gates_tstride; const float* C; size_t state_tstride;      // forward histories
    float* dcat; size_t dcat_tstride; int ld_cat;           // d[out | ctx | emb] of the output layer, [T][B][3H]; Tv > 5: the ctx block comes back as the TOTAL d(ctx_t)
    float* deh;                                             // Tv > 5: [T][Tv][B] floats of scratch (d(score) per step, for the dP / dV accumulation behind the launch)
    float* dZ; size_t dz_tstride;                           // [T][B][4H]
    const float* hWa; size_t hwa_tstride; const float* P; const float* Vt; const float* w; const float* alpha;
    const float* reg_coef; const float* asum; float reg_m;  // alpha regulariser (reg_coef NULL = none)
    float* dhWa; size_t dhwa_tstride;                       // [T][B][H], slots >= 1 written
    float* dP; float* dVt; float* dw;                       // accumulated (zeroed / owned by the caller)
    int B, H, T, Tv;
    float keep; uint32_t seed_lo, seed_hi, drop_code0; const int32_t* video_id; const int32_t* sample_id;
    float* img; float* ex; float* dctxs; unsigned* sync;    // scratch, sizes from attn_bwd_chain_scratch (16-byte aligned)
};
bool attn_bwd_chain_eligible(int B, int H, int Tv);
void attn_bwd_chain_scratch(int H, size_t* img_floats, size_t* ex_floats, size_t* row_floats, size_t* sync_bytes);
hipError_t launch_attn_bwd_chain(const AttnBwdChainLaunch& a, hipStream_t st);

// order-free NN contraction for the backward data path with optional split-K slabs:
// slab s (blockIdx.y) holds the partial over its K range at C + s * slab_stride.
struct NnBwdArgs {
    const float* A; int lda;        // [M, K]
    const float* W; int ldw;        // [K, N]
    float* C; int ldc;              // [M, N] (x splits)
    int M, N, K;
    int splits; size_t slab_stride;
};
hipError_t launch_gemm_nn_bwd(const NnBwdArgs& a, hipStream_t st);

}  // namespace s2vt
