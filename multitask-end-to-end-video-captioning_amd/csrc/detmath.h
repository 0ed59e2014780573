// detmath.h -- fixed-sequence fp32 transcendental functions + Philox4x32-10 for the gfx950 kernels.
//
// These are the product's statement of the numeric contract in DESIGN.md §3: every operation is
// an IEEE fp32 add / mul / fma / correctly-rounded divide or an integer bit operation, in a fixed
// order, so the result is a pure function of the input bits.  (The CPU oracle carries its own
// independent copy; tests/test_gpu_math.py compares the two bit-for-bit.)  Built with
// -ffp-contract=off: nothing here may be re-associated or contracted by the compiler.
//
// exp/log follow the Cephes single-precision schemes, tanh the clamped rational approximation
// published in Eigen (what TF-1.x evaluates on CPU for tf.tanh, used by BasicLSTMCell --
// reference tf_s2vt.py:74-77).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace s2vt {

__device__ __forceinline__ float dm_expf(float x)
{
    x = x < -87.0f ? -87.0f : x;
    x = x > 87.0f ? 87.0f : x;
    const float t = __builtin_fmaf(x, 1.44269504088896341f, 12582912.0f);
    const float n = t - 12582912.0f;
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = __builtin_fmaf(p, r, 1.3981999507E-3f);
    p = __builtin_fmaf(p, r, 8.3334519073E-3f);
    p = __builtin_fmaf(p, r, 4.1665795894E-2f);
    p = __builtin_fmaf(p, r, 1.6666665459E-1f);
    p = __builtin_fmaf(p, r, 5.0000001201E-1f);
    const float rr = r * r;
    float y = __builtin_fmaf(p, rr, r);
    y = y + 1.0f;
    const int ni = (int)n;
    return y * __uint_as_float((uint32_t)(ni + 127) << 23);
}

// exp(x) over the WHOLE fp32 range: same reduction and polynomial as dm_expf, no input clamp, the 2^n scaling applied in
// two halves so that the result overflows to +inf / underflows to 0 exactly where IEEE fp32 does (x >= 128 ln 2 =
// 88.7228...).  Only the reference's unshifted softmax (tf_s2vt.py:208-209, `exp(l) / sum(exp(l))`) needs this: there an
// overflowing logit turns into inf / inf = NaN.
__device__ __forceinline__ float dm_expf_ieee(float x)
{
    if (!(x < 89.0f)) return x != x ? x : __uint_as_float(0x7f800000u);
    if (x < -104.0f) return 0.0f;
    const float t = __builtin_fmaf(x, 1.44269504088896341f, 12582912.0f);
    const float n = t - 12582912.0f;
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = __builtin_fmaf(p, r, 1.3981999507E-3f);
    p = __builtin_fmaf(p, r, 8.3334519073E-3f);
    p = __builtin_fmaf(p, r, 4.1665795894E-2f);
    p = __builtin_fmaf(p, r, 1.6666665459E-1f);
    p = __builtin_fmaf(p, r, 5.0000001201E-1f);
    const float rr = r * r;
    float y = __builtin_fmaf(p, rr, r);
    y = y + 1.0f;
    const int ni = (int)n;
    const int n1 = ni / 2, n2 = ni - n1;                        // |n1|, |n2| <= 76: both factors are normal numbers
    return (y * __uint_as_float((uint32_t)(n1 + 127) << 23)) * __uint_as_float((uint32_t)(n2 + 127) << 23);
}

__device__ __forceinline__ float dm_logf(float x)
{
    const uint32_t b = __float_as_uint(x);
    int e = (int)((b >> 23) & 0xffu) - 126;
    float m = __uint_as_float((b & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m; }
    m = m - 1.0f;
    const float z = m * m;
    float p = 7.0376836292E-2f;
    p = __builtin_fmaf(p, m, -1.1514610310E-1f);
    p = __builtin_fmaf(p, m, 1.1676998740E-1f);
    p = __builtin_fmaf(p, m, -1.2420140846E-1f);
    p = __builtin_fmaf(p, m, 1.4249322787E-1f);
    p = __builtin_fmaf(p, m, -1.6668057665E-1f);
    p = __builtin_fmaf(p, m, 2.0000714765E-1f);
    p = __builtin_fmaf(p, m, -2.4999993993E-1f);
    p = __builtin_fmaf(p, m, 3.3333331174E-1f);
    float y = (p * m) * z;
    const float fe = (float)e;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(-0.5f, z, y);
    float r = m + y;
    r = __builtin_fmaf(fe, 0.693359375f, r);
    return r;
}

__device__ __forceinline__ float dm_tanhf(float x)
{
    x = x < -9.0f ? -9.0f : x;
    x = x > 9.0f ? 9.0f : x;
    const float x2 = x * x;
    float p = -2.76076847742355e-16f;
    p = __builtin_fmaf(x2, p, 2.00018790482477e-13f);
    p = __builtin_fmaf(x2, p, -8.60467152213735e-11f);
    p = __builtin_fmaf(x2, p, 5.12229709037114e-08f);
    p = __builtin_fmaf(x2, p, 1.48572235717979e-05f);
    p = __builtin_fmaf(x2, p, 6.37261928875436e-04f);
    p = __builtin_fmaf(x2, p, 4.89352455891786e-03f);
    p = x * p;
    float q = 1.19825839466702e-06f;
    q = __builtin_fmaf(x2, q, 1.18534705686654e-04f);
    q = __builtin_fmaf(x2, q, 2.26843463243900e-03f);
    q = __builtin_fmaf(x2, q, 4.89352518554385e-03f);
    return p / q;
}

__device__ __forceinline__ float dm_sigmoidf(float x) { return 1.0f / (1.0f + dm_expf(-x)); }

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11), counter-based: every noise word is a pure function of
// (key, counter), so any lane can produce any word and 1 GPU x B=64 draws the same numbers as
// 8 GPUs x B=8 (the counters carry GLOBAL video / sample indices).
// ---------------------------------------------------------------------------------------------
struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

__device__ __forceinline__ uint32_t pick_word(const u32x4& v, uint32_t i)
{
    return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
}

// uniform in (0,1): (k + 0.5) * 2^-23 with k the top 23 bits -- exactly representable
__device__ __forceinline__ float u01(uint32_t x)
{
    return __builtin_fmaf((float)(x >> 9), 1.1920928955078125e-07f, 5.9604644775390625e-08f);
}

// Gumbel(0,1) word of the sampler stream: counter (n>>2, video, sample, step), lane n&3
__device__ __forceinline__ float gumbel_at(uint32_t seed_lo, uint32_t seed_hi, uint32_t video, uint32_t sample,
                                           uint32_t step, uint32_t n)
{
    const u32x4 v = philox4x32_10(n >> 2, video, sample, step, seed_lo, seed_hi);
    const float u = u01(pick_word(v, n & 3u));
    return -dm_logf(-dm_logf(u));
}

// Gumbel words of FOUR consecutive columns 4q .. 4q+3 come from ONE Philox block, and in the MFMA accumulator
// layout those columns sit in four adjacent lanes (a quad) that also share their four rows.  So per (row
// group, column tile) quad lane e computes the block of row e only, and the 4x4 (row x column) words are
// exchanged inside the quad with DPP broadcasts: a quarter of the Philox work of calling gumbel_at per
// element, same words bit for bit.
template <int C>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, C * 0x55, 0xF, 0xF, true);    // quad_perm: every lane <- lane C
}
template <int R>
__device__ __forceinline__ uint32_t quad_word_from(const u32x4& mine, uint32_t e)
{
    const uint32_t x = quad_bcast<R>(mine.x), y = quad_bcast<R>(mine.y), z = quad_bcast<R>(mine.z), w = quad_bcast<R>(mine.w);
    return e == 0 ? x : e == 1 ? y : e == 2 ? z : w;
}
__device__ __forceinline__ float gumbel_from_word(uint32_t word) { return -dm_logf(-dm_logf(u01(word))); }
// The same value to ~1e-5 absolute from two v_log_f32 (1 ulp log2): only ever used to SCREEN candidates for the exact
// expression above (gemm_mfma.h, PICK epilogue) -- never decides a token.  u in [2^-24, 1 - 2^-24] keeps both logs normal.
constexpr float kGumbelScreenMargin = 1.0e-3f;       // >= 50 x the worst |fast - exact| + the rounding of logit + noise
__device__ __forceinline__ float gumbel_fast_from_word(uint32_t word)
{
    const float l = -0.693147180559945f * __builtin_amdgcn_logf(u01(word));
    return -0.693147180559945f * __builtin_amdgcn_logf(l);
}

// DropoutWrapper keep decision (reference tf_s2vt.py:75,77: floor(keep + U[0,1)) ) from the dropout
// stream: key (seed_lo, seed_hi ^ 'DROP'), counter (unit>>2, video, sample, code), code =
// layer*256 + unrolled step index.
__device__ __forceinline__ float dropout_keep01(uint32_t seed_lo, uint32_t seed_hi, uint32_t video, uint32_t sample,
                                                uint32_t code, uint32_t unit, float keep)
{
    const u32x4 v = philox4x32_10(unit >> 2, video, sample, code, seed_lo, seed_hi ^ 0x44524F50u);
    const float u = u01(pick_word(v, unit & 3u));
    return (keep + u) >= 1.0f ? 1.0f : 0.0f;
}

// total order on floats as unsigned integers (for the packed atomicMax argmax)
__device__ __forceinline__ uint32_t orderable(float f)
{
    const uint32_t b = __float_as_uint(f);
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}

}  // namespace s2vt
