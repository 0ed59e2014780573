/*
 * s2vt.h -- C ABI of libs2vt_hip.so: the MI355X-native S2VT REINFORCE hot path.
 *
 * The reference (adwardlee/multitask-end-to-end-video-captioning) has no FFI layer: its boundary
 * is the Python class Video_Caption_Generator whose build_* methods emit TensorFlow-1.1 graph ops
 * (tf_s2vt.py:53-266, reinforcement_multisampling_tf_s2vt.py:63-466).  Each entry point below
 * replaces the TF ops of one stretch of those graphs; the reference lines are cited per function.
 * INTEGRATION.md shows the ctypes binding and how Video_Caption_Generator.build_* map onto it.
 *
 * Conventions
 *   - Plain C: pointers, sizes, POD structs.  No torch / TF types.  All tensors are row-major
 *     fp32 (token ids int32), DEVICE pointers owned by the caller; the library allocates nothing
 *     on the device -- scratch comes from a caller-provided workspace (query *_workspace_bytes).
 *   - Every call takes a hipStream_t (as void*) and is asynchronous w.r.t. the host; no hidden
 *     synchronisation.  The core API is handle-free (all state is in the arguments); the session
 *     API at the end wraps the samplers around a handle that owns its workspace.
 *   - Return value: 0 = S2VT_OK, < 0 = S2VT_E_*.  Never throws, never exits.  After
 *     S2VT_E_HIP, s2vt_last_hip_error() holds the hipError_t.
 *   - Numeric contract: forward contractions are ascending-k fp32 fmaf chains on
 *     v_mfma_f32_16x16x4_f32, transcendental functions are fixed instruction sequences, sampling
 *     is Gumbel-max over Philox4x32-10 -- forward activations, logits and token ids are
 *     bit-identical to oracle/s2vt_oracle.c.  Gradients / reductions are order-free fp32.
 *   - Fast path: pointers 16-byte aligned, leading dimensions and K / N multiples of 4.  Matrices may be any size
 *     (operands are addressed per tile); a gather table passed as rowidx / token ids, a broadcast block (rowmod)
 *     and the weight matrix must each fit a 2 GiB window.  Anything else takes a scalar path with identical results.
 */
#ifndef S2VT_H
#define S2VT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S2VT_OK 0
#define S2VT_E_BADARG (-1)   /* null pointer, non-positive size, inconsistent dims */
#define S2VT_E_ALIGN (-2)    /* workspace not 256-byte aligned */
#define S2VT_E_WORKSPACE (-3)/* workspace too small */
#define S2VT_E_HIP (-4)      /* a HIP call failed: see s2vt_last_hip_error() */
#define S2VT_E_CHAIN_TIMEOUT (-5) /* a persistent recurrence timed out earlier (starved of CUs): activations since are suspect, variable
                                     updates queued behind it were SKIPPED on the device; see s2vt_chain_fault / s2vt_chain_ack */

typedef void* s2vt_stream;   /* hipStream_t */

/* Model dimensions: the constructor arguments of Video_Caption_Generator (tf_s2vt.py:54-66). */
typedef struct s2vt_dims {
    int32_t dim_image;            /* d   = 1536 */
    int32_t n_words;              /* |V|        */
    int32_t word_dim;             /* E   = 500  */
    int32_t lstm_dim;             /* H   = 1000 */
    int32_t n_video_lstm_step;    /* Tv  = 5    */
    int32_t n_caption_lstm_step;  /* Tc  = 20 (35 in the reference files) */
    int32_t label_dim;            /* A   = 400 attributes (multitask head), 0 if absent */
    int32_t reserved;
} s2vt_dims;

/* Trainable variables (tf_s2vt.py:68-88; LSTM kernels are created lazily by BasicLSTMCell under
 * s2vt/LSTM{1,2}/basic_lstm_cell/{weights,biases}).  The same struct carries gradients. */
typedef struct s2vt_params {
    float* Wemb;            /* [V, E]                                                    */
    float* encode_image_W;  /* [d, E]                                                    */
    float* encode_image_b;  /* [E]                                                       */
    float* lstm1_W;         /* [E + H, 4H]   rows [x ; h], columns [i | j | f | o]       */
    float* lstm1_b;         /* [4H]                                                      */
    float* lstm2_W;         /* [H + E + H, 4H] rows [out1 ; embed ; h2]                  */
    float* lstm2_b;         /* [4H]                                                      */
    float* embed_word_W;    /* [H, V]                                                    */
    float* embed_word_b;    /* [V]                                                       */
    float* attr_W;          /* [d, A] or NULL (reinforce_multitask_e2e_attribute_loss.py:112-114) */
    float* attr_b;          /* [A]    or NULL                                            */
} s2vt_params;

/* ---- library info / errors --------------------------------------------------------------- */
int s2vt_version(void);
/* bit 0: built with EXPERIMENTAL=1 (the opt-in decode-loop experiments decode4.hip / decode_loop.hip are linked in). */
int s2vt_build_flags(void);
int s2vt_last_hip_error(void);
/* Up to 8 device buffers set to zero by ONE launch (4-byte aligned, byte counts multiples of 4): the gradient bucket and the
 * other buffers a step must find zeroed -- each hipMemsetAsync / tensor-library fill is a ~5-20 us launch of its own. */
int s2vt_zero_regions(void* const* ptrs, const size_t* bytes, int32_t count, s2vt_stream stream);
const char* s2vt_error_string(int code);

/* ---- launch profiler (measurement only; bench.py's roofline leg) ---------------------------------
 * While enabled, every contraction launch is bracketed by two hipEvents recorded on the stream the
 * kernel is launched on and tallied per (kernel class, tile configuration) with its flops
 * (2*M*K*N of that launch, structural-zero segments excluded).  Call s2vt_prof_collect only after
 * synchronising the stream(s); it returns the number of rows written and resets the tally.
 * kernel_class: 0 = STORE contraction, 1 = fused LSTM cell, 2 = vocab logits + pick, 3 = TN weight grad. */
typedef struct s2vt_prof_row {
    int32_t kernel_class;
    int32_t tile_cfg;
    int64_t launches;
    double total_ms;
    double total_flops;
    char name[32];
} s2vt_prof_row;
int s2vt_prof_enable(int on);
/* Restrict the event brackets to one (kernel_class, tile_cfg); -1 = any.  Two events per launch cost ~1.5 us of
 * stream time each, ~10 % of a 600-launch step: bench.py profiles every launch during warm-up to find the dominant
 * kernel and only that kernel inside the timed region. */
int s2vt_prof_filter(int kernel_class, int tile_cfg);
int s2vt_prof_collect(s2vt_prof_row* rows, int max_rows);

/* ---- test hook: evaluate the contract's scalar functions on the device --------------------
 * fn: 0 exp, 1 log, 2 tanh, 3 sigmoid.  y[i] = fn(x[i]).  (Bitwise comparison with the oracle.) */
int s2vt_math_eval(int fn, const float* x, float* y, int64_t n, s2vt_stream stream);
/* Gumbel noise words of the sampler stream for (video, sample, step), columns [0, V). */
int s2vt_gumbel_eval(uint64_t seed, int32_t video, int32_t sample, int32_t step, float* out, int32_t V,
                     s2vt_stream stream);

/* ---- tf.nn.xw_plus_b / tf.matmul over a concatenated operand -------------------------------
 * C[m, :] = act( chain over [A0[r0(m)] ; A1[r1(m)] ; A2[r2(m)]] @ W  (+ bias) ), the chain
 * optionally continuing from Cinit.  Replaces tf.nn.xw_plus_b (tf_s2vt.py:98,153) and the
 * tf.concat + matmul inside BasicLSTMCell.  A segment with ptr == NULL is an all-zero input
 * (the reference's `padding`) and is skipped; rowidx gathers rows (tf.nn.embedding_lookup). */
typedef struct s2vt_operand {
    const float* ptr;       /* [rows, ld] */
    const int32_t* rowidx;  /* optional [M] */
    int32_t ld;
    int32_t k;              /* columns of this segment = rows of W it multiplies */
    int32_t rowmod;         /* > 0: row(m) = m % rowmod before rowidx */
    int32_t reserved;
} s2vt_operand;

int s2vt_gemm(const s2vt_operand* segs, int32_t nseg, const float* W, int32_t ldw, const float* bias,
              const float* Cinit, int32_t ldcinit, float* C, int32_t ldc, int32_t M, int32_t N, int32_t act_tanh,
              int32_t tile_cfg, s2vt_stream stream);

/* The same contraction with the weight matrix given TRANSPOSED, Wt[N, K] (row n = output column n, K contiguous, row
 * stride ldw): C = act([seg0;seg1;seg2] @ Wt^T + bias).  This is how the backward data-gradient products read the
 * forward weights as they lie (dX = dZ @ W^T) -- no transposed copies.  Same ascending-k chain, bit-identical to
 * s2vt_gemm on the transposed matrix. */
int s2vt_gemm_nt(const s2vt_operand* segs, int32_t nseg, const float* Wt, int32_t ldw, const float* bias,
                 const float* Cinit, int32_t ldcinit, float* C, int32_t ldc, int32_t M, int32_t N, int32_t act_tanh,
                 int32_t tile_cfg, s2vt_stream stream);

/* The order-free form of the same product for GRADIENTS (dX = dZ @ W^T, compared within tolerance, never part of the
 * bit-exact forward): C[M, N] = A[M, K] @ Wt[N, K]^T with the reduction cut into `splits` K slabs that run as
 * independent tiles -- a product with a long K and few output tiles (dO2 at B = 64: 1280 x 1000 outputs, K = 12000)
 * otherwise leaves most CUs idle -- written to `slabs` [splits][M][N] (slab_floats floats) and summed into C (ldc == N).
 * splits = 0: the library's choice for the shape (1 = no slabs; then `slabs` may be NULL); tile_cfg < 0: its tile. */
int s2vt_gemm_nt_splitk(const float* A, int32_t lda, const float* Wt, int32_t ldw, float* C, int32_t ldc, int32_t M,
                        int32_t N, int32_t K, int32_t splits, int32_t tile_cfg, float* slabs, size_t slab_floats,
                        s2vt_stream stream);

/* ---- BasicLSTMCell + DropoutWrapper, one call (tf_s2vt.py:74-77,119-143) --------------------
 * z = [x0 ; x1 ; h_prev] @ W + b ; i,j,f,o = split(z) ; c' = c*sig(f+1) + sig(i)*tanh(j) ;
 * h' = tanh(c')*sig(o) ; out = keep<1 ? (h'/keep)*mask : h'  (mask from the dropout Philox
 * stream, keyed by video_id/sample_id/drop_code).  W rows are consumed in the order x0, x1, h.
 * x0/x1 may be NULL (zero input) -- their rows of W are skipped, h_prev multiplies the LAST H rows.
 * gates (optional, [M,4H]) receives sig(i) | tanh(j) | sig(f+1) | sig(o) for the backward pass. */
int s2vt_lstm_cell_fwd(const s2vt_operand* x0, const s2vt_operand* x1, const float* h_prev, const float* c_prev,
                       int32_t state_rowmod, const float* W, const float* b, float* c_new, float* h_new, float* out,
                       float* gates, int32_t M, int32_t H, float keep, uint64_t seed, const int32_t* video_id,
                       const int32_t* sample_id, uint32_t drop_code, int32_t tile_cfg, s2vt_stream stream);

/* ---- vocab logits + token pick, one decode step ---------------------------------------------
 * logits = out2 @ embed_word_W + embed_word_b (tf_s2vt.py:153); token = tf.argmax(logits,1) for
 * rows with sample_id < 0 (tf_s2vt.py:262) or one draw of tf.multinomial(log_softmax(logits),1)
 * (reinforcement_multisampling_tf_s2vt.py:333-336) as Gumbel-max otherwise.  The logits stay
 * on chip unless logits_out != NULL.  packed[m] must be zero on entry; on exit it holds
 * (orderable(key) << 32) | ~token.  tokens_out (optional) receives the int32 ids. */
int s2vt_vocab_pick(const float* out2, int32_t ld, const float* W, const float* b, int32_t M, int32_t H, int32_t V,
                    const int32_t* video_id, const int32_t* sample_id, int32_t step, uint64_t seed,
                    unsigned long long* packed, int32_t* tokens_out, float* logits_out, int32_t tile_cfg,
                    s2vt_stream stream);

/* ---- frame embedding (tf_s2vt.py:97-101): emb[B*Tv, E] = video[B*Tv, d] @ encode_image_W + b */
int s2vt_frame_embed_fwd(const s2vt_dims* d, const s2vt_params* p, const float* video, int32_t B, float* emb,
                         s2vt_stream stream);

/* ---- samplers: build_multinomial_sampler x K + build_sampler, one call ------------------------
 * (reinforcement_multisampling_tf_s2vt.py:294-391, driven at :743-753).  Frame embed + encode once,
 * then Tc decode steps for K sampled rows-blocks (+ one greedy block when with_greedy) entirely
 * on the device: no host round trip per step.  Rows are sample-major (row k*B + j = sample k of
 * video j, :764-782), the greedy block last.  ids_out: int32 [(K + with_greedy) * B, Tc].
 * Noise: Philox stream (seed; video = video_base + j; sample = k; step). */
size_t s2vt_sample_workspace_bytes(const s2vt_dims* d, int32_t B, int32_t K, int32_t with_greedy);
int s2vt_sample(const s2vt_dims* d, const s2vt_params* p, const float* video, int32_t B, int32_t K,
                int32_t with_greedy, uint64_t seed, int32_t video_base, int32_t* ids_out, void* workspace,
                size_t workspace_bytes, s2vt_stream stream);
/* s2vt_sample with flags.  S2VT_SAMPLE_STOP_AT_EOS (opt-in; NOT what the reference does, whose samplers run all Tc steps for every row,
 * reinforcement_multisampling_tf_s2vt.py:318-337): a row leaves the decode loop once it has picked <eos> = 0 -- every later step's cell
 * and vocabulary launches cover the rows still sampling only (compact row lists kept on the device, no host round trip) -- and its ids
 * behind the first <eos> are 0.  Ids up to and including the first <eos> are bit-identical to s2vt_sample's, and those are the only
 * positions the objective looks at (the mask of cider_evaluation.py:145-172 ends there). */
#define S2VT_SAMPLE_STOP_AT_EOS 1
int s2vt_sample_ex(const s2vt_dims* d, const s2vt_params* p, const float* video, int32_t B, int32_t K, int32_t with_greedy,
                   uint64_t seed, int32_t video_base, int32_t flags, int32_t* ids_out, void* workspace, size_t workspace_bytes,
                   s2vt_stream stream);

/* ---- teacher-forced unroll: build_model (tf_s2vt.py:90-153) / build_loss
 * (reinforcement_multisampling_tf_s2vt.py:227-292) forward --------------------------------------
 * video [B,Tv,d]; the N = rep*B rows are sample-major copies of the B videos (row n uses video
 * n % B -- the reference tiles the feature block 8x on the host, :779-782; here it is never
 * materialised).  caption int32 [N,Tc]; previous word = <bos> at t=0 else caption[:,t-1].
 * Dropout-wrapped cells: keep < 1 draws the masks from the dropout Philox stream (seed; video_id,
 * sample_id per row; code = layer*256 + step); keep >= 1 disables dropout.  logits_out
 * [Tc*N, V] is TIME-MAJOR (row t*N + n).  Activations needed by s2vt_bptt_bwd stay in the workspace. */
size_t s2vt_train_workspace_bytes(const s2vt_dims* d, int32_t B, int32_t N);
int s2vt_teacher_forced_fwd(const s2vt_dims* d, const s2vt_params* p, const float* video, int32_t B, int32_t N,
                            const int32_t* caption, float keep, uint64_t seed, const int32_t* video_id,
                            const int32_t* sample_id, float* logits_out, void* workspace, size_t workspace_bytes,
                            s2vt_stream stream);

/* Same, inside a REINFORCE step: LSTM1's state trajectory depends only on the frames and the weights, and the
 * sampler pass of the step (s2vt_sample / s2vt_encode_fwd on the SAME video block, SAME weights) has just
 * computed it.  Pass that call's workspace (and its row count R = (K + with_greedy) * B) and the trajectory and
 * gates are copied from it instead of being recomputed.  sampler_workspace == NULL: identical to the above. */
int s2vt_teacher_forced_fwd_reuse(const s2vt_dims* d, const s2vt_params* p, const float* video, int32_t B, int32_t N,
                                  const int32_t* caption, float keep, uint64_t seed, const int32_t* video_id,
                                  const int32_t* sample_id, float* logits_out, void* workspace, size_t workspace_bytes,
                                  const void* sampler_workspace, size_t sampler_workspace_bytes, int32_t sampler_rows,
                                  s2vt_stream stream);

/* Same, unrolling only the first `caption_steps` (1 .. Tc) decode steps: when no row of the batch has an unmasked
 * position at t >= caption_steps -- the padding behind the longest caption of the batch (tf_s2vt.py:371-401 pads every
 * caption to Tc; cider_evaluation.py:145-172 masks a sample behind its first <eos>) -- those steps add exact zeros to the
 * loss and to every gradient, so a caller that knows the masks on the host skips them.  `caption` keeps its [N, Tc] row
 * stride; logits_out is [caption_steps * N, V]; the workspace is the one s2vt_train_workspace_bytes sizes (its layout does
 * not depend on caption_steps).  Pair it with s2vt_bptt_bwd_steps at the SAME caption_steps. */
int s2vt_teacher_forced_fwd_steps(const s2vt_dims* d, const s2vt_params* p, const float* video, int32_t B, int32_t N,
                                  const int32_t* caption, int32_t caption_steps, float keep, uint64_t seed,
                                  const int32_t* video_id, const int32_t* sample_id, float* logits_out, void* workspace,
                                  size_t workspace_bytes, const void* sampler_workspace, size_t sampler_workspace_bytes,
                                  int32_t sampler_rows, s2vt_stream stream);

/* Same, and the vocabulary projection only for the LIVE (step, row) pairs: live_rows[r] (ascending, < caption_steps * N,
 * time-major index t * N + n) names the unrolled row whose logits land in row r of logits_out [n_live, V].  A position behind a
 * sample's first <eos> is masked (cider_evaluation.py:145-172): its logits feed nothing, its loss term and every gradient
 * contribution are exact zeros -- on a trained model's samples (8 of 20 positions live) the four vocabulary-sized kernels of a
 * step (logits, softmax, dWout, dO2: a third of it) shrink with the live fraction.  At 257-384 rows LSTM2's recurrence also stops
 * a row behind its last live step (rows sorted by length on the device; the history slots of a stopped row are not written and
 * nothing reads them); the other recurrence forms step every row.
 * live_rows == NULL (n_live == 0): every row, as s2vt_teacher_forced_fwd_steps.  Pair it with s2vt_bptt_bwd_live on the same list.
 * The list must be closed towards earlier steps (t * N + n live => (t - 1) * N + n live), as masks up to a first <eos> are: the
 * backward also leaves the dead rows out of LSTM2's weight- and input-gradient products, whose dZ2 rows are zeros only BEHIND
 * a row's last live step. */
int s2vt_teacher_forced_fwd_live(const s2vt_dims* d, const s2vt_params* p, const float* video, int32_t B, int32_t N,
                                 const int32_t* caption, int32_t caption_steps, const int32_t* live_rows, int32_t n_live, float keep,
                                 uint64_t seed, const int32_t* video_id, const int32_t* sample_id, float* logits_out,
                                 void* workspace, size_t workspace_bytes, const void* sampler_workspace,
                                 size_t sampler_workspace_bytes, int32_t sampler_rows, s2vt_stream stream);

/* ---- softmax / NLL rows, forward + backward ----------------------------------------------------
 * nll[r] = -sum_v q[v] * log_softmax(logits[r])[v],  q = onehot(target[r])*(1-s) + s/V
 * (tf.losses.softmax_cross_entropy(label_smoothing=s), tf_s2vt.py:155; s = 0 gives the
 * log-prob of the sampled word, reinforcement_multisampling_tf_s2vt.py:286-288), and
 * logits[r,:] <- coef[r] * (softmax(logits[r]) - q)   (the gradient of sum_r coef[r]*nll[r]).
 * XE: coef = the Q1 batch-mean weights; REINFORCE: coef[r] = (reward - baseline)[n] * mask[n,t]
 * (:643-646) -- in both cases WITHOUT the 1/sum(mask) factor, which is applied after the
 * data-parallel all-reduce by s2vt_grad_finalize.  lp_target (optional) = log-prob of target. */
int s2vt_softmax_nll_fwd_bwd(float* logits, int32_t ld, int32_t R, int32_t V, const int32_t* target, const float* coef,
                             float smoothing, float* nll, float* lp_target, s2vt_stream stream);
/* The same with one label-smoothing value PER ROW (smoothing_rows [R]): rows of two objectives -- the reward-scaled NLL of
 * sampled captions (s = 0) and the smoothed XE of ground-truth captions (s = 0.05), reinforce_multitask_e2e_attribute_s2vt.py:850
 * -- share one teacher-forced pass. */
int s2vt_softmax_nll_fwd_bwd_rows(float* logits, int32_t ld, int32_t R, int32_t V, const int32_t* target, const float* coef,
                                  const float* smoothing_rows, float* nll, float* lp_target, s2vt_stream stream);

/* ---- back-propagation through the unroll (what tf.gradients builds, :650) ---------------------
 * dlogits [Tc*N, V] time-major (output of s2vt_softmax_nll_fwd_bwd); the workspace must still hold
 * the activations of the matching s2vt_teacher_forced_fwd call (same d, B, N, keep, seed, ids).
 * Gradients are ACCUMULATED into `grads` (same layout as the parameters; zero it first). */
int s2vt_bptt_bwd(const s2vt_dims* d, const s2vt_params* p, const s2vt_params* grads, const float* video, int32_t B,
                  int32_t N, const float* dlogits, float keep, uint64_t seed, const int32_t* video_id,
                  const int32_t* sample_id, void* workspace, size_t workspace_bytes, s2vt_stream stream);

/* The same in parts, for data-parallel callers that start the all-reduce of a gradient slice as soon as it is final and
 * let it run under the rest of the backward:  phase 1 = the vocab projection (embed_word_W / _b final; also the
 * gradient w.r.t. LSTM2's outputs);  3 = LSTM2's recurrence and weight gradients (lstm2_W / _b final);  4 = everything
 * after (input gradients, LSTM1, Wemb, frame embedding);  2 = 3 + 4;  0 = s2vt_bptt_bwd.  Call them in the order
 * 1, 3, 4 (or 1, 2) on the same workspace. */
int s2vt_bptt_bwd_phase(const s2vt_dims* d, const s2vt_params* p, const s2vt_params* grads, const float* video, int32_t B,
                        int32_t N, const float* dlogits, float keep, uint64_t seed, const int32_t* video_id,
                        const int32_t* sample_id, void* workspace, size_t workspace_bytes, int32_t phase, s2vt_stream stream);

/* The backward of s2vt_teacher_forced_fwd_steps: dlogits is [caption_steps * N, V]; steps behind caption_steps carry no
 * gradient (the recurrences start from zero at the last unrolled step).  phase as above. */
int s2vt_bptt_bwd_steps(const s2vt_dims* d, const s2vt_params* p, const s2vt_params* grads, const float* video, int32_t B,
                        int32_t N, const float* dlogits, int32_t caption_steps, float keep, uint64_t seed,
                        const int32_t* video_id, const int32_t* sample_id, void* workspace, size_t workspace_bytes, int32_t phase,
                        s2vt_stream stream);

/* The backward of s2vt_teacher_forced_fwd_live: dlogits is [n_live, V] (row r = unrolled row live_rows[r]); the vocabulary
 * gradients are reduced over the live rows, the gradient w.r.t. LSTM2's outputs is computed for them and is zero elsewhere;
 * LSTM2's weight gradients and d[out1 ; embed] likewise take the Tv encode steps plus the live decode rows. */
int s2vt_bptt_bwd_live(const s2vt_dims* d, const s2vt_params* p, const s2vt_params* grads, const float* video, int32_t B,
                       int32_t N, const float* dlogits, int32_t caption_steps, const int32_t* live_rows, int32_t n_live, float keep,
                       uint64_t seed, const int32_t* video_id, const int32_t* sample_id, void* workspace, size_t workspace_bytes,
                       int32_t phase, s2vt_stream stream);

/* Gradient w.r.t. the frame features, for the end-to-end scripts where they are the CNN's output
 * (e2e_tf_s2vt.py:106-121,163-166: the optimizer differentiates through `video` into Inception-ResNet-v2):
 * d_video[B, Tv, dim_image] = d_emb @ encode_image_W^T, from the d_emb the preceding s2vt_bptt_bwd (phase 0 or 2)
 * left in `workspace`.  Unnormalised like the weight gradients (scale by 1/sum(mask) as s2vt_grad_finalize does). */
int s2vt_bptt_dvideo(const s2vt_dims* d, const s2vt_params* p, int32_t B, int32_t N, float* d_video, void* workspace,
                      size_t workspace_bytes, s2vt_stream stream);

/* dWemb[idx[r], :] += dE[r, :]  -- gradient of tf.nn.embedding_lookup (tf_s2vt.py:128-134). */
int s2vt_embed_scatter_add(const float* dE, int32_t ld, const int32_t* idx, int32_t R, int32_t E, float* dWemb,
                           s2vt_stream stream);

/* ---- host glue of the REINFORCE step, on the device (decode_captions_masks, cider_evaluation.py:145-172; the objective's
 * coefficients, reinforcement_multisampling_tf_s2vt.py:641-646).
 * s2vt_caption_mask: ids [N, Tc] -> mask [N, Tc] (1 up to and including the first <eos> = 0; may be NULL), target_tm [Tc*N]
 *   (the ids time-major, what s2vt_softmax_nll_fwd_bwd takes; may be NULL), *mask_sum = *mask_sum_copy = sum(mask) (each may be NULL).
 * s2vt_pg_coef: coef_tm[t*N + n] = mask[n][t] * (rewards[n] - baseline[n]) * scale  (rewards NULL = 1, baseline NULL = 0).
 * s2vt_step_scalars: *loss = sum(coef * nll) / *mask_sum_local, *gscale = 1 / *mask_sum_global (the bucket's tail slot after the
 *   all-reduce), *sumsq = 0 -- every output optional; one launch. */
int s2vt_caption_mask(const int32_t* ids, int32_t N, int32_t Tc, float* mask, int32_t* target_tm, float* mask_sum, float* mask_sum_copy, s2vt_stream stream);
int s2vt_pg_coef(const float* mask, const float* rewards, const float* baseline, float scale, int32_t N, int32_t Tc, float* coef_tm,
                 s2vt_stream stream);
int s2vt_step_scalars(const float* coef, const float* nll, int64_t R, const float* mask_sum_local, const float* mask_sum_global, float* loss,
                      float* gscale, float* sumsq, s2vt_stream stream);

/* ---- the same for the XE update (tf_s2vt.py:150-166) and the mixed multitask objective (reinforce_multitask_e2e_attribute_s2vt.py:850):
 * one launch each instead of the tensor-library expressions (single process; a data-parallel caller needs the GLOBAL column
 * sums and keeps its own expressions).  Tc <= 128.
 * s2vt_xe_prep: coef_tm[t*N + n] = q1 ? (sum_n' mask[n'][t] / n_global) * loss_weight : mask[n][t] * loss_weight;
 *   target_tm[t*N + n] = caption[n][t] (may be NULL); *mask_sum = sum(mask) (may be NULL).
 * s2vt_mixed_prep: rows 0..Ns-1 = the sampled captions, Ns..Ns+B-1 = the ground truth (N = Ns + B):
 *   coef_tm[t*N + n] = n < Ns ? mask[n][t] * ((rewards[n] - baseline[n]) * (1 - lambda)) / sum(mask)
 *                             : (q1 ? (sum_b gt_mask[b][t] / n_global_b) * loss_weight : gt_mask[n-Ns][t] * loss_weight) * (lambda / sum(gt_mask));
 *   smooth_tm[t*N + n] = n < Ns ? 0 : smoothing;  caption_all [N, Tc] = [sampled ; gt_caption], target_tm [Tc*N] the same ids time-major;
 *   sums = {sum(mask), sum(gt_mask)}.  lambda_loss is a DOUBLE: (1 - lambda) is formed in double and rounded to fp32 once, lambda itself
 *   rounded to fp32 -- what the tensor expressions of a data-parallel caller (model.mixed_update) do with a Python float, so the two
 *   paths give the same bits for a non-dyadic lambda (0.9: 0.1 -> 0x3dcccccd, not the 0x3dccccd0 of 1 - fl32(0.9)).
 * s2vt_mixed_loss: out3 = {sum over sampled rows, sum over ground-truth rows, both} of coef[r] * nll[r], r < R; the row of entry r is
 *   (live_rows ? live_rows[r] : r) % N. */
int s2vt_xe_prep(const float* mask, const int32_t* caption, int32_t N, int32_t Tc, float loss_weight, float n_global, int32_t q1, float* coef_tm,
                 int32_t* target_tm, float* mask_sum, s2vt_stream stream);
int s2vt_mixed_prep(const float* mask, const float* gt_mask, const float* rewards, const float* baseline, const int32_t* sampled,
                    const int32_t* gt_caption, int32_t Ns, int32_t B, int32_t Tc, double lambda_loss, float loss_weight, int32_t q1,
                    float smoothing, float n_global_b, float* coef_tm, float* smooth_tm, int32_t* caption_all, int32_t* target_tm, float* sums, s2vt_stream stream);
int s2vt_mixed_loss(const float* coef, const float* nll, const int32_t* live_rows, int64_t R, int32_t N, int32_t Ns, float* out3, s2vt_stream stream);

/* ---- gradient finalisation + tf.clip_by_global_norm + tf.train.AdamOptimizer ------------------
 * (reinforcement_multisampling_tf_s2vt.py:638-652; tf_s2vt.py:163-166,445-448).
 * s2vt_grad_finalize: g <- g * (*gscale) + weight_decay * theta over one flat range, and
 *   *sumsq += sum g^2.  gscale (device scalar, may be NULL = 1) carries 1/sum(mask) computed after
 *   the all-reduce; weight_decay implements decay_value * l2_loss for the decayed variables (Q3).
 * s2vt_adam_tf: g' = g * clip_norm / max(sqrt(*sumsq), clip_norm) (clip_norm <= 0 or sumsq NULL:
 *   no clipping), then TF-form Adam with step count `step` >= 1 (epsilon outside the bias
 *   correction). */
int s2vt_grad_finalize(float* g, const float* theta, int64_t n, const float* gscale, float weight_decay, float* sumsq,
                       s2vt_stream stream);
int s2vt_adam_tf(float* theta, const float* g, float* m, float* v, int64_t n, const float* sumsq, float clip_norm,
                 float lr, int64_t step, float beta1, float beta2, float eps, s2vt_stream stream);
/* The same update with a receipt: when the launch applies the update it writes `step` to *applied_step (device int32).
 * Every s2vt_adam_tf* launch is a no-op ON THE DEVICE while a persistent-recurrence fault is pending (s2vt_chain_fault):
 * gradients computed from a starved recurrence never reach the variables, and after a device synchronise
 * *applied_step tells the host which update was the last one applied. */
int s2vt_adam_tf_guarded(float* theta, const float* g, float* m, float* v, int64_t n, const float* sumsq, float clip_norm,
                         float lr, int64_t step, float beta1, float beta2, float eps, int32_t* applied_step, s2vt_stream stream);

/* ---- temporal attention, one decode step (original_attention.py:106-128) ----------------------
 * Inputs: hWa [B,H] = h_prev @ embed_att_Wa (s2vt_gemm); P [Tv,B,H] = Vt @ embed_att_Ua + ba (hoisted,
 * :107); Vt [Tv,B,H] frame embeddings; w [H] = embed_att_w.
 *   scores[t,b] = sum_h tanh(hWa[b,h] + P[t,b,h]) * w[h]            (:113-115)
 *   alpha[t,b]  = exp(scores) / (sum_t exp(scores) (+1 if that sum is 0))   (:116-121, no max shift)
 *   ctx[b,h]    = sum_t alpha[t,b] * Vt[t,b,h]                       (:127-128)
 * Backward: given dctx [B,H] returns dhWa [B,H], dP [Tv,B,H], dVt [Tv,B,H] (the context path only;
 * add dP @ Ua^T upstream) and ACCUMULATES dw [H].  de_scratch: [Tv*B] floats. */
int s2vt_attention_fwd(const float* hWa, const float* P, const float* Vt, const float* w, float* scores, float* alpha,
                       float* ctx, int32_t Tv, int32_t B, int32_t H, s2vt_stream stream);
int s2vt_attention_bwd(const float* hWa, const float* P, const float* Vt, const float* w, const float* alpha,
                       const float* dctx, float* de_scratch, float* dhWa, float* dP, float* dVt, float* dw, int32_t Tv,
                       int32_t B, int32_t H, s2vt_stream stream);

/* ---- the temporal-attention captioner as a whole model (original_attention.py:53-251) -----------
 * Variables of original_attention.py:55-86 in the reference's own layouts (names = the TF variable names; LSTM3 lives under
 * s2vt/LSTM3/basic_lstm_cell).  The word embedding has dim_hidden columns (:65); s2vt_dims.word_dim is ignored here.
 * Numeric contract: every pre-activation is one ascending-k fmaf chain over the rows of its matrix, the
 * row blocks in order of availability -- LSTM3: current_embed [H:2H], h [2H:3H], atten [0:H]; output layer: current_embed
 * [2H:3H], atten [H:2H], output1 [0:H]; forward activations, alphas, logits and greedy ids are bit-identical to
 * oracle/s2vt_oracle.py::attention_forward.  n_video_lstm_step <= 64. */
typedef struct s2vt_attn_params {
    float* Wemb;            /* [V, H] */
    float* encode_image_W;  /* [D, H] */
    float* encode_image_b;  /* [H] */
    float* embed_att_w;     /* [H] (TF shape [H, 1]) */
    float* embed_att_Wa;    /* [H, H] */
    float* embed_att_Ua;    /* [H, H] */
    float* embed_att_ba;    /* [H] */
    float* embed_word_W;    /* [H, V] */
    float* embed_word_b;    /* [V] */
    float* embed_nn_Wp;     /* [3H, H]  rows [output1 ; atten ; current_embed]  (:134) */
    float* embed_nn_bp;     /* [H] */
    float* lstm3_W;         /* [3H, 4H] rows [atten ; current_embed ; h], columns [i | j | f | o]  (:131) */
    float* lstm3_b;         /* [4H] */
} s2vt_attn_params;

size_t s2vt_attn_workspace_bytes(const s2vt_dims* d, int32_t B);
/* build_model's unroll (:88-143) on B rows with the DropoutWrapper on LSTM3's output (keep, Philox stream as the S2VT cells:
 * code 768 + step): logits [caption_steps*B, V] time-major (row t*B + b); caption [B, Tc] int32 (the word fed at step t is
 * caption[:, t-1], zeros at t = 0, :105,141-142).  caption_steps (1..Tc): unroll only the leading steps (every later position of
 * the batch is masked).  alphas_out: optional [caption_steps][Tv][B].  Activations stay in the workspace for s2vt_attn_bptt_bwd. */
int s2vt_attn_teacher_forced_fwd(const s2vt_dims* d, const s2vt_attn_params* p, const float* video, int32_t B, const int32_t* caption,
                                 int32_t caption_steps, float keep, uint64_t seed, const int32_t* video_id, const int32_t* sample_id,
                                 float* logits, float* alphas_out, void* workspace, size_t workspace_bytes, s2vt_stream stream);
/* The loss kernels' inputs from what build_model is fed: caption [B, Tc] int32 and caption_mask [B, Tc] (row-major) -> time-major
 * target_tm[t*B + b], coef_tm[t*B + b] = mask[b, t] (cross_entropy * caption_mask[:, i], :145), reg_tm = beta * mask (NULL: not wanted),
 * *mask_sum = sum(mask) (:149).  One launch. */
int s2vt_attn_loss_inputs(const int32_t* caption, const float* mask, int32_t B, int32_t Tc, float beta, int32_t* target_tm, float* coef_tm,
                          float* reg_tm, float* mask_sum, s2vt_stream stream);
/* loss = (sum_r coef[r] * nll[r] + sum_r reg_coef[r] * max(0, reg_m - sum(alpha[0:8])[r])) / *mask_sum_local  (:144-149; reg_coef =
 * beta * mask time-major or NULL, reg_m = m; the first-8-frames sums are the ones the forward left in the workspace);
 * *gscale = 1 / *mask_sum_global; *sumsq = 0.  One launch, after s2vt_attn_teacher_forced_fwd on the same workspace. */
int s2vt_attn_step_scalars(const float* coef, const float* nll, int64_t R, const float* reg_coef, float reg_m, const float* mask_sum_local,
                           const float* mask_sum_global, float* loss, float* gscale, float* sumsq, const s2vt_dims* d, int32_t B,
                           void* workspace, size_t workspace_bytes, s2vt_stream stream);
/* tf.gradients of (sum coef * nll + sum regulariser) through the unroll the forward call left in the workspace: dlogits
 * [caption_steps*B, V] = d/dlogits (s2vt_softmax_nll_fwd_bwd), reg_coef [caption_steps*B] = beta * mask[b, t] time-major or NULL.
 * ACCUMULATES into `grads` (same layout as the variables; un-normalised: scale by 1 / sum(mask) afterwards).  lstm_dim % 4 == 0. */
int s2vt_attn_bptt_bwd(const s2vt_dims* d, const s2vt_attn_params* p, const s2vt_attn_params* grads, const float* video, int32_t B,
                       const float* dlogits, int32_t caption_steps, const float* reg_coef, float reg_m, float keep, uint64_t seed,
                       const int32_t* video_id, const int32_t* sample_id, void* workspace, size_t workspace_bytes, s2vt_stream stream);
/* build_generator / build_sampler (:155-251): greedy decode of B videos, the whole loop on the device (the picked word of step
 * t-1 is the gather index of step t's embedding operand); ids_out [B, Tc] int32, alphas_out optional [Tc][Tv][B] (saved_alphas). */
int s2vt_attn_decode_greedy(const s2vt_dims* d, const s2vt_attn_params* p, const float* video, int32_t B, int32_t video_base, int32_t* ids_out,
                            float* alphas_out, void* workspace, size_t workspace_bytes, s2vt_stream stream);

/* ---- multitask attribute head (reinforce_multitask_e2e_attribute_loss.py:375-380, 606-626) -----
 * mean_feat[b,:] = mean_t video[b,t,:]; z = mean_feat @ attr_W + attr_b;
 * bce = max(z,0) - z*y + log(1+exp(-|z|))  (tf.nn.sigmoid_cross_entropy_with_logits); labels/bce optional.
 * Backward: dz = scale * (sigmoid(z) - y); d_attr_W += mean_feat^T dz; d_attr_b += colsum(dz)
 * (scale = alpha / (A * B_global) for the normalised multilabel loss at :379,957). */
int s2vt_attr_head_fwd(const float* video, int32_t B, int32_t Tv, int32_t D, const float* attr_W, const float* attr_b,
                       int32_t A, const float* labels, float* mean_feat, float* z, float* bce, s2vt_stream stream);
int s2vt_attr_head_bwd(const float* mean_feat, const float* z, const float* labels, int32_t B, int32_t D, int32_t A,
                       float scale, float* dz_scratch, float* d_attr_W, float* d_attr_b, s2vt_stream stream);
/* evaluate_multilabel (reinforce_multitask_e2e_attribute_loss.py:606-626): the head's forward followed by
 * scores[B, A] = sigmoid(z) (:624).  mean_feat [B, D] and z [B, A] are written as by s2vt_attr_head_fwd. */
int s2vt_attr_head_scores(const float* video, int32_t B, int32_t Tv, int32_t D, const float* attr_W, const float* attr_b,
                          int32_t A, float* mean_feat, float* z, float* scores, s2vt_stream stream);

/* ==== session API + the single-op entry points of the boundary ============= */

/* ---- session handle: owns ONE device workspace sized for (max_B, max_K); nothing else is allocated.
 * s2vt_encode_fwd = encoding stage (tf_s2vt.py:97-122) + the decode-stage work that does not depend on a
 * sampled word (the LSTM1 trajectory and its W2 partials); s2vt_decode_greedy = build_sampler
 * (tf_s2vt.py:217-266; B = 1 is build_generator :169-214); s2vt_decode_multinomial = K runs of
 * build_multinomial_sampler (reinforcement_multisampling_tf_s2vt.py:294-339, driven at :743-753) on the
 * SAME encode -- any number of decode calls may follow one encode call.  ids_out: int32 [B, Tc] resp.
 * [K*B, Tc] (sample-major).  One handle per (device, stream); a handle is not thread-safe. */
typedef struct s2vt_handle s2vt_handle;
int s2vt_create(const s2vt_dims* d, int32_t max_B, int32_t max_K, s2vt_handle** out);
int s2vt_destroy(s2vt_handle* h);
int s2vt_encode_fwd(s2vt_handle* h, const s2vt_params* p, const float* video, int32_t B, s2vt_stream stream);
int s2vt_decode_greedy(s2vt_handle* h, const s2vt_params* p, int32_t* ids_out, s2vt_stream stream);
int s2vt_decode_multinomial(s2vt_handle* h, const s2vt_params* p, int32_t K, uint64_t seed, int32_t video_base,
                            int32_t* ids_out, s2vt_stream stream);

/* ---- BasicLSTMCell weight layouts.  The kernels consume the TF layout directly (W[in+H, 4H], columns
 * [i | j | f | o], rows [x ; h]); these convert to / from a split, gate-interleaved form
 * (Wx[in, H, 4], Wh[H, H, 4]: the four gates of a unit adjacent) for hosts that keep i2h / h2h apart. */
int s2vt_pack_weights(const float* W_tf, int32_t in_dim, int32_t H, float* Wx_packed, float* Wh_packed, s2vt_stream stream);
int s2vt_unpack_weights(const float* Wx_packed, const float* Wh_packed, int32_t in_dim, int32_t H, float* W_tf,
                        s2vt_stream stream);

/* ---- frame embedding backward (gradient of tf.nn.xw_plus_b, tf_s2vt.py:98): ACCUMULATES
 * d_encode_image_W[d, E] += video[B*Tv, d]^T @ d_emb[B*Tv, E] and d_encode_image_b[E] += colsum(d_emb). */
int s2vt_frame_embed_bwd(const s2vt_dims* d, const float* video, const float* d_emb, int32_t B, float* d_encode_image_W,
                         float* d_encode_image_b, s2vt_stream stream);

/* ---- BasicLSTMCell backward, one step (inverse of s2vt_lstm_cell_fwd's pointwise part): gates [M,4H] as
 * saved by the forward, dh [M,H] = total gradient w.r.t. h' (recurrent + output), dc_in [M,H] or NULL,
 * c_prev NULL = zero state.  Writes dz [M,4H] (pre-activation gate gradients, columns i|j|f|o) and
 * dc_prev [M,H]; the weight / input gradients are contractions with dz (s2vt_gemm / s2vt_bptt_bwd). */
int s2vt_lstm_cell_bwd(const float* gates, const float* c_new, const float* c_prev, const float* dh, const float* dc_in,
                       float* dz, float* dc_prev, int32_t M, int32_t H, s2vt_stream stream);

/* ---- build_generator's word choice exactly as the reference writes it (tf_s2vt.py:208-209): p = exp(l) / sum(exp(l))
 * WITHOUT a max shift, in fp32, then argmax (first maximum wins; NaN never wins; all-NaN -> 0).  A logit >= 88.72
 * overflows to inf / inf = NaN and the choice becomes <eos> = 0 instead of argmax(l).  ids [R]; probs [R, V] or NULL. */
int s2vt_softmax_unshifted_argmax(const float* logits, int32_t ld, int32_t R, int32_t V, int32_t* ids, float* probs,
                                  s2vt_stream stream);

/* ---- the two objectives by name.  s2vt_xent_smooth_fwd_bwd: tf.losses.softmax_cross_entropy with
 * label_smoothing (tf_s2vt.py:155).  s2vt_pg_nll_fwd_bwd: the reward-scaled NLL of
 * reinforcement_multisampling_tf_s2vt.py:286-291,643-646 -- coef[t*N+n] = adv[n] * mask[n,t] is formed on
 * the device (coef_scratch: N*Tc floats), logits/target time-major; both overwrite logits with d/dlogits. */
int s2vt_xent_smooth_fwd_bwd(float* logits, int32_t ld, int32_t R, int32_t V, const int32_t* target, const float* coef,
                             float label_smoothing, float* nll, s2vt_stream stream);
int s2vt_pg_nll_fwd_bwd(float* logits, int32_t ld, int32_t N, int32_t Tc, int32_t V, const int32_t* target_tm,
                        const float* adv, const float* mask, float* coef_scratch, float* nll, float* lp_target,
                        s2vt_stream stream);

/* out[r, :] = Wemb[idx[r], :]  -- tf.nn.embedding_lookup (tf_s2vt.py:128-134). */
int s2vt_embed_gather(const float* Wemb, int32_t ldw, const int32_t* idx, int32_t R, int32_t E, float* out, int32_t ldo,
                      s2vt_stream stream);

/* tf.clip_by_global_norm over one flat range (reinforcement_multisampling_tf_s2vt.py:650):
 * g *= clip_norm / max(||g||, clip_norm); *sumsq_scratch receives ||g||^2 (before clipping). */
int s2vt_global_norm_clip(float* g, int64_t n, float clip_norm, float* sumsq_scratch, s2vt_stream stream);

/* ---- generic order-free pieces for backward graphs composed on the host (used by the attention captioner's
 * training step, attention.py; the S2VT path has them fused inside s2vt_bptt_bwd) ---------------------------
 * s2vt_gemm_tn : C[Kout,N] (+)= A[row(m), :Kout]^T @ B[m, :N] over m < Mred  (weight gradients; rowidx gathers rows)
 * s2vt_transpose: out[c, r] = in[r, c]                       s2vt_colsum: out[n] += sum_m X[m, n]  (bias gradients)
 * s2vt_tanh_bwd : dx = dy * (1 - y^2)                        s2vt_dropout_bwd: dh = (dout / keep) * mask, the adjoint
 *                 of the DropoutWrapper output of s2vt_lstm_cell_fwd (same seed / ids / drop_code). */
int s2vt_gemm_tn(const float* A, int32_t lda, const int32_t* rowidx, const float* B, int32_t ldb, float* C, int32_t ldc,
                 int32_t Mred, int32_t Kout, int32_t N, int32_t accumulate, s2vt_stream stream);
int s2vt_transpose(const float* in, int32_t ldi, float* out, int32_t ldo, int32_t R, int32_t Cc, s2vt_stream stream);
int s2vt_colsum(const float* X, int32_t ld, int32_t M, int32_t N, float* out, s2vt_stream stream);
int s2vt_tanh_bwd(const float* y, const float* dy, float* dx, int64_t n, s2vt_stream stream);
int s2vt_dropout_bwd(const float* dout, int32_t ld, float* dh, int32_t M, int32_t H, float keep, uint64_t seed,
                     uint32_t drop_code, const int32_t* video_id, const int32_t* sample_id, s2vt_stream stream);

/* ---- a whole BasicLSTMCell recurrence in one call: the unroll of tf_s2vt.py:113-153 for a cell whose step input is
 * known before the loop.  Step t = 0 .. T-1:  z = cinit_t (+) H_hist[t] @ W[kw0 : kw0 + H, :]  (cinit_t = cinit +
 * t * cinit_tstride, rows ldcinit apart, for t < cinit_steps; absent otherwise), BasicLSTMCell pointwise with C_hist[t],
 * results to C_hist[t+1], H_hist[t+1] ([T+1, M, H], slot 0 = the initial state), the activated gates to gates[t]
 * ([T, M, 4H] or NULL; may alias cinit) and the DropoutWrapper output to out[t] ([T, M, H] or NULL; Philox code
 * drop_code0 + t).  persistent = 1: ONE persistent launch with the recurrent weights resident in LDS and the state
 * exchanged through L2 (needs M <= 384, H % 4 == 0, H <= 1024, H / 4 <= the CU count; S2VT_E_BADARG otherwise); 0: T launches of
 * the fused cell kernel; -1: persistent when the shape fits.  Both forms give the same bits.  scratch:
 * s2vt_lstm_recurrence_scratch_bytes(H) bytes, 256-byte aligned (persistent form only). */
size_t s2vt_lstm_recurrence_scratch_bytes(int32_t H);
int s2vt_lstm_recurrence_fwd(const float* W, int32_t kw0, const float* b, const float* cinit, int64_t cinit_tstride, int32_t ldcinit,
                             int32_t cinit_steps, float* C_hist, float* H_hist, float* gates, float* out, int32_t M, int32_t H,
                             int32_t T, float keep, uint64_t seed, const int32_t* video_id, const int32_t* sample_id,
                             uint32_t drop_code0, int32_t persistent, void* scratch, size_t scratch_bytes, s2vt_stream stream);
/* ---- back-propagation through that recurrence (tf.gradients through the unroll, reinforcement_multisampling_tf_s2vt.py:650):
 * for t = T-1 .. 0:  dh = dext_t (through the DropoutWrapper mask, Philox code drop_code0 + t; dext_t = dext + (t - dext_t0) *
 * dext_tstride, rows ld_ext apart, for t >= dext_t0; dext NULL = none) + dZ[t+1] @ W[kw0 : kw0 + H, :]^T;  dZ[t] ([T, M, 4H],
 * pre-activation gradients in the i | j | f | o column order) from dh, the carried cell gradient, gates[t] ([T, M, 4H]
 * activated, as s2vt_lstm_recurrence_fwd saved them) and C_hist ([T+1, M, H]).  persistent = 1: ONE launch -- workgroup
 * (unit group, gate) keeps its [H x 16] slice of the recurrent rows in LDS, dz crosses the chip as per-gate images in MFMA
 * operand order, the four gate partials of a unit group meet through a 4-workgroup exchange, the cell gradient stays in
 * registers (M <= 384, H % 4 == 0, H <= 1024, 4 * ceil(H / 16) <= the CU count; S2VT_E_BADARG otherwise; above 256 rows a
 * workgroup owns 32 units and half the row tiles); 0: per step one pointwise launch + split-K slabs of the product; -1:
 * persistent when the shape fits (M <= 384).  Results agree to
 * reduction order (gradients are order-free).  scratch: s2vt_lstm_recurrence_bwd_scratch_bytes(M, H)
 * bytes, 256-byte aligned. */
size_t s2vt_lstm_recurrence_bwd_scratch_bytes(int32_t M, int32_t H);
int s2vt_lstm_recurrence_bwd(const float* W, int32_t kw0, const float* gates, const float* C_hist, const float* dext, int64_t dext_tstride,
                             int32_t ld_ext, int32_t dext_t0, float* dZ, int32_t M, int32_t H, int32_t T, float keep, uint64_t seed,
                             const int32_t* video_id, const int32_t* sample_id, uint32_t drop_code0, int32_t persistent, void* scratch,
                             size_t scratch_bytes, s2vt_stream stream);
/* Grid-wide waits of the persistent recurrence that gave up (bounded spins), over all launches of this process; 0 =
 * healthy.  Read after synchronising the stream. */
int s2vt_chain_timeouts(void);
/* The persistent recurrence needs all its workgroups resident at once.  The launcher checks the occupancy and never
 * overlaps two persistent grids of one process, but it cannot see another process on the same GPU: if a grid-wide wait
 * gives up, a fault is raised (host-visible counter + device-resident word) and stays raised until acknowledged:
 *   - s2vt_chain_fault(): 1 while a fault is pending (a host-memory read: no synchronisation, callable every step);
 *   - every entry point that launches a recurrence or updates variables returns S2VT_E_CHAIN_TIMEOUT while it is pending,
 *     and s2vt_adam_tf* launches already queued behind the faulting kernel skip their update on the device;
 *   - s2vt_chain_ack(disable): synchronises the device, clears the fault; disable != 0 turns the persistent form off for
 *     the rest of the process (the per-step launches give the same bits).  The caller then repeats the work since the
 *     last applied update. */
int s2vt_chain_fault(void);
int s2vt_chain_ack(int disable_persistent);
/* s2vt_chain_hold(1) ... s2vt_chain_hold(0) (nestable): while held, every recurrence whose form the library chooses takes its
 * per-step launches (same bits, ~5 % slower) instead of the persistent grid.  For callers that have OTHER kernels in flight on
 * the same GPU beside the library's stream -- an asynchronous RCCL all-reduce of a gradient slice (model.backward(overlap=True)):
 * a persistent grid needs every workgroup co-resident and must not be started while part of the chip is taken. */
int s2vt_chain_hold(int on);

/* In-place SUM all-reduce of the flat gradient bucket over an existing RCCL communicator (ncclComm_t as
 * void*), on `stream` -- the exchange step of SURVEY.md section 8(e) for C/C++ hosts (Python hosts use
 * torch.distributed).  The library never loads RCCL itself (a communicator belongs to the instance that made it):
 * it calls, in this order, the entry point registered with s2vt_set_rccl_allreduce, a `ncclAllReduce` visible in the
 * global symbol scope, or one from an already loaded librccl; S2VT_E_BADARG if there is none. */
int s2vt_allreduce_grads(float* bucket, int64_t n, void* rccl_comm, s2vt_stream stream);
/* Register the `ncclAllReduce` of the host's own RCCL (needed when it was loaded RTLD_LOCAL); NULL unregisters. */
int s2vt_set_rccl_allreduce(void* nccl_allreduce_fn_ptr);

#ifdef __cplusplus
}
#endif
#endif /* S2VT_H */
