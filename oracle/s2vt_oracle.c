/*
 * s2vt_oracle.c -- CPU restatement (TEST INFRASTRUCTURE, not product code) of the
 * arithmetic on the S2VT REINFORCE hot path of adwardlee/multitask-end-to-end-video-captioning.
 *
 * PARITY STATUS: "parity unpinned" by the reference.  The reference is Python-2 /
 * TensorFlow-1.1 graph code (tf_s2vt.py, reinforcement_multisampling_tf_s2vt.py,
 * original_attention.py, reinforce_multitask_e2e_attribute_loss.py); TensorFlow 1.1.0 is
 * an un-vendored third-party dependency that cannot be installed here, and the reference
 * holds no tests or golden vectors.  What this file restates is therefore the published
 * semantics of the TF-1.1 ops the reference calls (cited per function).  The only part of
 * the reference that could be executed here (its TF-free host helpers) pins
 * tests/golden/hostglue.json; see tools/make_fixtures.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * NUMERIC CONTRACT (what "bit-exact" means for the HIP path):
 *   - every contraction is ONE fp32 fused-multiply-add chain per output element, over
 *     ascending k, starting from +0 (or from a carried partial): acc = fmaf(a[k], w[k], acc).
 *     gfx950's v_mfma_f32_16x16x4_f32 / 32x32x2_f32 are bit-for-bit such chains
 *     (MI355X guide, "FP32-input MFMA"), so a kernel that walks k in ascending order
 *     without split-K reproduces these numbers exactly.
 *   - exp / log / tanh / sigmoid are the fixed instruction sequences below (IEEE fp32
 *     add, mul, fma, correctly rounded divide, integer bit ops only), following the
 *     published Cephes single-precision kernels that Eigen (TF's CPU math) also uses.
 *   - sampling is Gumbel-max over Philox4x32-10 counters (tf.multinomial's GPU kernel
 *     is Gumbel-max; its Philox stream is not reproducible without TF, so the counter
 *     scheme here is this project's own and is documented in DESIGN.md).
 *
 * Build: see oracle/Makefile (gcc -O2 -mfma -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------
 * Deterministic transcendental functions
 * ---------------------------------------------------------------------------------- */
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* exp(x), Cephes expf scheme (range reduction by ln2 hi/lo, degree-5 polynomial).
 * Input clamped to [-87, 87] so that the result and 1/(1+result) stay normal numbers. */
static inline float det_expf(float x)
{
    x = x < -87.0f ? -87.0f : x;
    x = x > 87.0f ? 87.0f : x;
    const float t = fmaf(x, 1.44269504088896341f, 12582912.0f); /* round-to-nearest-even */
    const float n = t - 12582912.0f;
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = fmaf(p, r, 1.3981999507E-3f);
    p = fmaf(p, r, 8.3334519073E-3f);
    p = fmaf(p, r, 4.1665795894E-2f);
    p = fmaf(p, r, 1.6666665459E-1f);
    p = fmaf(p, r, 5.0000001201E-1f);
    const float rr = r * r;
    float y = fmaf(p, rr, r);
    y = y + 1.0f;
    const int32_t ni = (int32_t)n;
    return y * u2f((uint32_t)(ni + 127) << 23);
}

/* exp(x) over the whole fp32 range (no clamp; 2^n applied in two halves so overflow / underflow happen where IEEE fp32
 * puts them): used only by the reference's UNSHIFTED softmax, tf_s2vt.py:208-209. */
static inline float det_expf_ieee(float x)
{
    if (!(x < 89.0f)) return x != x ? x : u2f(0x7f800000u);
    if (x < -104.0f) return 0.0f;
    const float t = fmaf(x, 1.44269504088896341f, 12582912.0f);
    const float n = t - 12582912.0f;
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = fmaf(p, r, 1.3981999507E-3f);
    p = fmaf(p, r, 8.3334519073E-3f);
    p = fmaf(p, r, 4.1665795894E-2f);
    p = fmaf(p, r, 1.6666665459E-1f);
    p = fmaf(p, r, 5.0000001201E-1f);
    const float rr = r * r;
    float y = fmaf(p, rr, r);
    y = y + 1.0f;
    const int32_t ni = (int32_t)n;
    const int32_t n1 = ni / 2, n2 = ni - n1;
    return (y * u2f((uint32_t)(n1 + 127) << 23)) * u2f((uint32_t)(n2 + 127) << 23);
}

/* log(x) for normal x > 0, Cephes logf scheme. */
static inline float det_logf(float x)
{
    const uint32_t b = f2u(x);
    int32_t e = (int32_t)((b >> 23) & 0xffu) - 126;
    float m = u2f((b & 0x007fffffu) | 0x3f000000u); /* [0.5, 1) */
    if (m < 0.707106781186547524f) { e -= 1; m = m + m; }
    m = m - 1.0f;
    const float z = m * m;
    float p = 7.0376836292E-2f;
    p = fmaf(p, m, -1.1514610310E-1f);
    p = fmaf(p, m, 1.1676998740E-1f);
    p = fmaf(p, m, -1.2420140846E-1f);
    p = fmaf(p, m, 1.4249322787E-1f);
    p = fmaf(p, m, -1.6668057665E-1f);
    p = fmaf(p, m, 2.0000714765E-1f);
    p = fmaf(p, m, -2.4999993993E-1f);
    p = fmaf(p, m, 3.3333331174E-1f);
    float y = (p * m) * z;
    const float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(-0.5f, z, y);
    float r = m + y;
    r = fmaf(fe, 0.693359375f, r);
    return r;
}

/* tanh(x): the clamped rational approximation (odd degree-13 / even degree-6) published in
 * Eigen's generic_fast_tanh_float, which is what TF-1.x CPU tanh evaluates. */
static inline float det_tanhf(float x)
{
    x = x < -9.0f ? -9.0f : x;
    x = x > 9.0f ? 9.0f : x;
    const float x2 = x * x;
    float p = -2.76076847742355e-16f;
    p = fmaf(x2, p, 2.00018790482477e-13f);
    p = fmaf(x2, p, -8.60467152213735e-11f);
    p = fmaf(x2, p, 5.12229709037114e-08f);
    p = fmaf(x2, p, 1.48572235717979e-05f);
    p = fmaf(x2, p, 6.37261928875436e-04f);
    p = fmaf(x2, p, 4.89352455891786e-03f);
    p = x * p;
    float q = 1.19825839466702e-06f;
    q = fmaf(x2, q, 1.18534705686654e-04f);
    q = fmaf(x2, q, 2.26843463243900e-03f);
    q = fmaf(x2, q, 4.89352518554385e-03f);
    return p / q;
}

static inline float det_sigmoidf(float x) { return 1.0f / (1.0f + det_expf(-x)); }

ORC_API void orc_expf(const float* x, float* y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = det_expf(x[i]); }
ORC_API void orc_logf(const float* x, float* y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = det_logf(x[i]); }
ORC_API void orc_tanhf(const float* x, float* y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = det_tanhf(x[i]); }
ORC_API void orc_sigmoidf(const float* x, float* y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = det_sigmoidf(x[i]); }

/* ------------------------------------------------------------------------------------
 * tf.nn.xw_plus_b / tf.matmul: one ascending-k fmaf chain per output.
 *   C[m, n] = chain_k( A[row(m), k] * W[k, n] )  starting from (accumulate ? C[m,n] : +0)
 * row(m) = rowidx ? rowidx[m] : m      (tf.nn.embedding_lookup folded into the operand,
 *                                       tf_s2vt.py:128-134)
 * Bias is added afterwards by orc_bias_add (xw_plus_b = matmul, then add).
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_gemm_chain(const float* A, int64_t lda, const int32_t* rowidx,
                            const float* W, int64_t ldw, float* C, int64_t ldc,
                            int64_t M, int64_t K, int64_t N, int accumulate)
{
    enum { RB = 4, CB = 256 };
    const int64_t nrb = (M + RB - 1) / RB, ncb = (N + CB - 1) / CB;
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t rb = 0; rb < nrb; ++rb) {
        for (int64_t cb = 0; cb < ncb; ++cb) {
            float acc[RB][CB];
            const int64_t m0 = rb * RB, n0 = cb * CB;
            const int64_t mr = (M - m0 < RB) ? (M - m0) : RB;
            const int64_t nc = (N - n0 < CB) ? (N - n0) : CB;
            const float* arow[RB];
            for (int64_t r = 0; r < RB; ++r) {
                const int64_t m = m0 + (r < mr ? r : 0);
                const int64_t src = rowidx ? (int64_t)rowidx[m] : m;
                arow[r] = A + src * lda;
                for (int64_t j = 0; j < nc; ++j)
                    acc[r][j] = (accumulate && r < mr) ? C[(m0 + r) * ldc + n0 + j] : 0.0f;
            }
            for (int64_t k = 0; k < K; ++k) {
                const float* w = W + k * ldw + n0;
                const float a0 = arow[0][k], a1 = arow[1][k], a2 = arow[2][k], a3 = arow[3][k];
                for (int64_t j = 0; j < nc; ++j) {
                    const float wv = w[j];
                    acc[0][j] = fmaf(a0, wv, acc[0][j]);
                    acc[1][j] = fmaf(a1, wv, acc[1][j]);
                    acc[2][j] = fmaf(a2, wv, acc[2][j]);
                    acc[3][j] = fmaf(a3, wv, acc[3][j]);
                }
            }
            for (int64_t r = 0; r < mr; ++r)
                for (int64_t j = 0; j < nc; ++j) C[(m0 + r) * ldc + n0 + j] = acc[r][j];
        }
    }
}

ORC_API void orc_bias_add(float* C, int64_t ldc, const float* b, int64_t M, int64_t N)
{
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m)
        for (int64_t n = 0; n < N; ++n) C[m * ldc + n] = C[m * ldc + n] + b[n];
}

/* ------------------------------------------------------------------------------------
 * BasicLSTMCell pointwise part (TF 1.1 tf.contrib.rnn.BasicLSTMCell, forget_bias=1.0,
 * state_is_tuple=False; used at tf_s2vt.py:119,122,140,143 -- SURVEY App. B2):
 *   z = [i | j | f | o]  (already includes the bias)
 *   c' = c * sigmoid(f + 1) + sigmoid(i) * tanh(j) ;  h' = tanh(c') * sigmoid(o)
 * DropoutWrapper(output_keep_prob=keep) (tf_s2vt.py:75,77 -- App. B3):
 *   out = (h' / keep) * mask, state not dropped.  mask==NULL -> out = h' (samplers).
 * gates_out (optional, [M,4H]) receives the activated gates [si | tj | sf | so].
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_lstm_pointwise(const float* z, const float* c_prev, float* c_new, float* h_new,
                                float* out, const float* drop_mask, float keep, float* gates_out,
                                int64_t M, int64_t H)
{
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        const float* zr = z + m * 4 * H;
        for (int64_t u = 0; u < H; ++u) {
            const float si = det_sigmoidf(zr[u]);
            const float tj = det_tanhf(zr[H + u]);
            const float sf = det_sigmoidf(zr[2 * H + u] + 1.0f);
            const float so = det_sigmoidf(zr[3 * H + u]);
            const float t1 = c_prev[m * H + u] * sf;
            const float t2 = si * tj;
            const float c = t1 + t2;
            const float h = det_tanhf(c) * so;
            c_new[m * H + u] = c;
            h_new[m * H + u] = h;
            if (out) out[m * H + u] = drop_mask ? (h / keep) * drop_mask[m * H + u] : h;
            if (gates_out) {
                float* g = gates_out + m * 4 * H;
                g[u] = si; g[H + u] = tj; g[2 * H + u] = sf; g[3 * H + u] = so;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------
 * Philox4x32-10 (Salmon et al., SC'11) -- counter-based noise for the sampler.
 * ---------------------------------------------------------------------------------- */
static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

ORC_API void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c[4] = { ctr[0], ctr[1], ctr[2], ctr[3] };
    philox4x32_10(c, key[0], key[1]);
    memcpy(out, c, sizeof(c));
}

/* uniform in (0,1) from 23 random bits, exactly representable: (k + 0.5) * 2^-23 */
static inline float u01(uint32_t x) { return fmaf((float)(x >> 9), 1.1920928955078125e-07f, 5.9604644775390625e-08f); }

/* Gumbel(0,1) noise word for (video v, sample s, step t, vocab column n).
 * counter = (n >> 2, v, s, t), key = (seed_lo, seed_hi), lane = n & 3. */
static inline float gumbel_at(uint32_t seed_lo, uint32_t seed_hi, uint32_t v, uint32_t s, uint32_t t, uint32_t n)
{
    uint32_t c[4] = { n >> 2, v, s, t };
    philox4x32_10(c, seed_lo, seed_hi);
    const float u = u01(c[n & 3]);
    return -det_logf(-det_logf(u));
}

/* DropoutWrapper keep decision floor(keep + u) (tf_s2vt.py:75,77) from the dropout
 * stream: key (seed_lo, seed_hi ^ 'DROP'), counter (unit >> 2, video, sample, code), lane unit & 3;
 * code = layer * 256 + unrolled step index.  out[m*H + u] in {0, 1}. */
ORC_API void orc_dropout_mask(uint32_t seed_lo, uint32_t seed_hi, const int32_t* video_id, const int32_t* sample_id,
                              uint32_t code, float keep, float* out, int64_t M, int64_t H)
{
    for (int64_t m = 0; m < M; ++m)
        for (int64_t u = 0; u < H; ++u) {
            uint32_t c[4] = { (uint32_t)u >> 2, (uint32_t)video_id[m], (uint32_t)sample_id[m], code };
            philox4x32_10(c, seed_lo, seed_hi ^ 0x44524F50u);
            out[m * H + u] = (keep + u01(c[u & 3])) >= 1.0f ? 1.0f : 0.0f;
        }
}

ORC_API void orc_gumbel_noise(uint32_t seed_lo, uint32_t seed_hi, uint32_t v, uint32_t s, uint32_t t,
                              float* out, int64_t V)
{
    for (int64_t n = 0; n < V; ++n) out[n] = gumbel_at(seed_lo, seed_hi, v, s, t, (uint32_t)n);
}

/* ------------------------------------------------------------------------------------
 * Token pick for one decode step (logits already include the bias).
 *   greedy : tf.argmax(logit_words, 1) -- lowest index among ties
 *            (tf_s2vt.py:262, reinforcement_multisampling_tf_s2vt.py:387)
 *   sample : tf.multinomial(tf.nn.log_softmax(logits), 1)
 *            (reinforcement_multisampling_tf_s2vt.py:333-336): one categorical draw from
 *            softmax(logits); restated as Gumbel-max: argmax_n(logit[n] + g[n]).  The
 *            log_softmax shift is a per-row constant and cannot change the argmax.
 * video_id[m], sample_id[m] name the noise stream of row m; sample_id < 0 -> greedy row.
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_pick_tokens(const float* logits, int64_t ldl, int64_t M, int64_t V,
                             const int32_t* video_id, const int32_t* sample_id, int32_t step,
                             uint32_t seed_lo, uint32_t seed_hi, int32_t* tok_out)
{
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t m = 0; m < M; ++m) {
        const float* l = logits + m * ldl;
        const int greedy = sample_id[m] < 0;
        float best = -INFINITY; int32_t bi = 0;
        for (int64_t n = 0; n < V; ++n) {
            float key = l[n];
            if (!greedy)
                key = key + gumbel_at(seed_lo, seed_hi, (uint32_t)video_id[m], (uint32_t)sample_id[m],
                                      (uint32_t)step, (uint32_t)n);
            if (key > best || n == 0) { best = key; bi = (int32_t)n; }
        }
        tok_out[m] = bi;
    }
}

/* ------------------------------------------------------------------------------------
 * Row losses on logits [M, V] (bias included):
 *   lse   = max + log(sum_n exp(l[n] - max))      (tf.nn.log_softmax, App. B6)
 *   q     = onehot * (1 - s) + s / V              (label_smoothing, App. B5; s = 0 for PG)
 *   nll   = - sum_n q[n] * (l[n] - lse)
 * The sum over n is sequential ascending (fp32); the HIP kernel reduces in a different
 * order, so these values are compared within a tolerance, never bitwise.
 * Also returns lp_target = l[target] - lse (the one non-zero of the dense
 * softmax_value*onehot tensor, reinforcement_multisampling_tf_s2vt.py:286-288).
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_row_losses(const float* logits, int64_t ldl, int64_t M, int64_t V,
                            const int32_t* target, float smoothing, float* nll, float* lp_target,
                            float* lse_out)
{
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        const float* l = logits + m * ldl;
        float mx = l[0];
        for (int64_t n = 1; n < V; ++n) mx = l[n] > mx ? l[n] : mx;
        float s = 0.0f;
        for (int64_t n = 0; n < V; ++n) s = s + det_expf(l[n] - mx);
        const float lse = mx + det_logf(s);
        const float qoff = smoothing / (float)V;
        const float qon = (1.0f - smoothing) + qoff;
        float acc = 0.0f;
        if (smoothing != 0.0f) {
            for (int64_t n = 0; n < V; ++n) {
                const float lp = l[n] - lse;
                acc = fmaf(n == target[m] ? qon : qoff, lp, acc);
            }
        } else {
            acc = l[target[m]] - lse;
        }
        if (nll) nll[m] = -acc;
        if (lp_target) lp_target[m] = l[target[m]] - lse;
        if (lse_out) lse_out[m] = lse;
    }
}

/* ------------------------------------------------------------------------------------
 * build_generator's word choice as written (tf_s2vt.py:208-209): probs = exp(l) / reduce_sum(exp(l)) with NO max shift,
 * then tf.argmax.  fp32; an overflowing logit gives inf / inf = NaN; argmax keeps the FIRST maximum and a NaN never
 * compares greater, so an all-NaN / NaN-and-zeros row yields index 0.  TF's reduce_sum has no defined order; the
 * numeric contract ("reductions that decide a token") gives it one -- 256 strided partial sums
 * (element v into partial v mod 256, ascending), an xor butterfly (32, 16, .. 1) inside each group of 64 partials, then
 * ((g0 + g1) + g2) + g3 -- and this function and the product kernel both implement that definition.
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_softmax_unshifted_argmax(const float* logits, int64_t ldl, int64_t M, int64_t V, int32_t* ids, float* probs)
{
    for (int64_t m = 0; m < M; ++m) {
        const float* l = logits + m * ldl;
        float part[256];
        for (int t = 0; t < 256; ++t) {
            float s = 0.0f;
            for (int64_t i = t; i < V; i += 256) s = s + det_expf_ieee(l[i]);
            part[t] = s;
        }
        float grp[4];
        for (int g = 0; g < 4; ++g) {
            float a[64], b[64];
            for (int i = 0; i < 64; ++i) a[i] = part[g * 64 + i];
            for (int o = 32; o > 0; o >>= 1) {
                for (int i = 0; i < 64; ++i) b[i] = a[i] + a[i ^ o];
                for (int i = 0; i < 64; ++i) a[i] = b[i];
            }
            grp[g] = a[0];
        }
        const float total = ((grp[0] + grp[1]) + grp[2]) + grp[3];
        int64_t best = 0;
        float bestv = -3.402823466e+38f;
        for (int64_t i = 0; i < V; ++i) {
            const float pv = det_expf_ieee(l[i]) / total;
            if (probs) probs[m * V + i] = pv;
            if (pv > bestv) { bestv = pv; best = i; }
        }
        ids[m] = (int32_t)best;
    }
}

/* ------------------------------------------------------------------------------------
 * Temporal attention score / softmax / context for one decode step
 * (original_attention.py:113-128):
 *   e[t,b]  = sum_h tanh(hWa[b,h] + P[t,b,h]) * w[h]        (ascending-h fmaf chain)
 *   a[t,b]  = exp(e[t,b]) / (sum_t exp(e[t,b]) (+1 if that sum == 0))   (no max shift)
 *   ctx[b,h]= sum_t a[t,b] * Vemb[t,b,h]                    (ascending-t fmaf chain)
 * hWa = h_prev @ Wa is computed by orc_gemm_chain; P = Vemb @ Ua + ba is hoisted (:107).
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_attention_step(const float* hWa, const float* P, const float* Vemb, const float* w,
                                float* alpha /*[Tv,B]*/, float* ctx /*[B,H]*/,
                                int64_t Tv, int64_t B, int64_t H)
{
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        float ex[64];
        float den = 0.0f;
        for (int64_t t = 0; t < Tv; ++t) {
            const float* p = P + (t * B + b) * H;
            float e = 0.0f;
            for (int64_t h = 0; h < H; ++h) e = fmaf(det_tanhf(hWa[b * H + h] + p[h]), w[h], e);
            ex[t] = det_expf(e);
            den = den + ex[t];
        }
        if (den == 0.0f) den = den + 1.0f;
        for (int64_t t = 0; t < Tv; ++t) alpha[t * B + b] = ex[t] / den;
        for (int64_t h = 0; h < H; ++h) {
            float c = 0.0f;
            for (int64_t t = 0; t < Tv; ++t) c = fmaf(alpha[t * B + b], Vemb[(t * B + b) * H + h], c);
            ctx[b * H + h] = c;
        }
    }
}

ORC_API void orc_tanh_inplace(float* x, int64_t n)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) x[i] = det_tanhf(x[i]);
}

/* ------------------------------------------------------------------------------------
 * Attribute head (reinforce_multitask_e2e_attribute_loss.py:375-380):
 *   a = mean_t video[b,t,:]  (sum ascending t, then divide by Tv)
 *   z = a @ attr_W + attr_b  (orc_gemm_chain + orc_bias_add)
 *   bce = max(z,0) - z*y + log1p(exp(-|z|))   (tf.nn.sigmoid_cross_entropy_with_logits, B11)
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_mean_frames(const float* video, float* out, int64_t B, int64_t Tv, int64_t D)
{
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b)
        for (int64_t d = 0; d < D; ++d) {
            float s = 0.0f;
            for (int64_t t = 0; t < Tv; ++t) s = s + video[(b * Tv + t) * D + d];
            out[b * D + d] = s / (float)Tv;
        }
}

ORC_API void orc_sigmoid_bce(const float* z, const float* y, float* bce, int64_t n)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float zz = z[i];
        const float az = zz < 0.0f ? -zz : zz;
        const float sp = det_logf(1.0f + det_expf(-az));
        bce[i] = ((zz > 0.0f ? zz : 0.0f) - zz * y[i]) + sp;
    }
}

ORC_API int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
