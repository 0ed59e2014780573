// frag_gemm.hip -- feasibility probe (diagnostic only): a contraction whose operands already lie in MFMA FRAGMENT ORDER in
// global memory, so the main loop has no LDS, no barrier, no staging stores: every wave streams its own A / B fragments
// (one fully coalesced 1 KB load per 16-row x 16-k block) into a register ring and issues v_mfma_f32_16x16x4_f32.
//   A image: [row tile][k group][lane = (k%4)*16 + row%16][component = (k%16)/4]      (what chain.hip exchanges for h)
//   B image: [k group][col tile][lane = (k%4)*16 + col%16][component = (k%16)/4]
// Shapes of the bench: PICK (384 x 1000 x 12000) and the logits product (6400 x 1000 x 12000).  Prints TFLOP/s per tiling.
//   hipcc --offload-arch=gfx950 -O3 -o frag_gemm tools/micro/frag_gemm.hip && ./frag_gemm
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } \
    } while (0)

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// WM x WN waves per workgroup, TM x TN 16x16 accumulators per wave, RING k-groups in flight
template <int WM, int WN, int TM, int TN, int RING>
__global__ __launch_bounds__(64 * WM * WN) void frag_gemm(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                          int mt, int nt, int kg, int ldc)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware order as the product's skinny-M mapping: all row tiles of a column tile on one XCD
    const int ntile_n = nt / (WN * TN), ntile_m = mt / (WM * TM);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile_m = slot % ntile_m, tile_n = (slot / ntile_m) * 8 + xcd;
    if (tile_n >= ntile_n) return;
    const int rt0 = (tile_m * WM + wm) * TM, ct0 = (tile_n * WN + wn) * TN;
    const f32x4* Ap = reinterpret_cast<const f32x4*>(A) + lane;
    const f32x4* Bp = reinterpret_cast<const f32x4*>(B) + lane;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ra[RING][TM], rb[RING][TN];
    auto issue = [&](int g, f32x4 (&qa)[TM], f32x4 (&qb)[TN]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TM; ++i) qa[i] = Ap[((size_t)(rt0 + i) * kg + g) * 64];
#pragma unroll
        for (int j = 0; j < TN; ++j) qb[j] = Bp[((size_t)g * nt + ct0 + j) * 64];
    };
    static_for<0, RING>([&](auto r_) { constexpr int r = decltype(r_)::value; issue(r < kg ? r : kg - 1, ra[r], rb[r]); });
    __builtin_amdgcn_sched_barrier(0);
    for (int g0 = 0; g0 < kg; g0 += RING) {
        static_for<0, RING>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            const int g = g0 + r;
            if (g < kg) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[r][i][e], rb[r][j][e], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int gn = g + RING < kg ? g + RING : kg - 1;
                issue(gn, ra[r], rb[r]);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    }
    // epilogue: plain stores (row-major C) so that the result can be checked
    const int l15 = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(size_t)((rt0 + i) * 16 + lq * 4 + r) * ldc + (ct0 + j) * 16 + l15] = acc[i][j][r];
}

static void pack_a(const std::vector<float>& src, int M, int K, int kg, std::vector<float>& dst)
{
    dst.assign((size_t)(M / 16) * kg * 256, 0.f);
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k)
            dst[((size_t)(m / 16) * kg + k / 16) * 256 + ((k % 4) * 16 + m % 16) * 4 + (k % 16) / 4] = src[(size_t)m * K + k];
}
static void pack_b(const std::vector<float>& src, int K, int N, int kg, std::vector<float>& dst)
{
    dst.assign((size_t)kg * (N / 16) * 256, 0.f);
    for (int k = 0; k < K; ++k)
        for (int n = 0; n < N; ++n)
            dst[((size_t)(k / 16) * (N / 16) + n / 16) * 256 + ((k % 4) * 16 + n % 16) * 4 + (k % 16) / 4] = src[(size_t)k * N + n];
}

template <int WM, int WN, int TM, int TN, int RING>
void run(const char* name, int M, int K, int N, const float* dA, const float* dB, float* dC, const std::vector<float>& hA, const std::vector<float>& hB)
{
    const int kg = (K + 15) / 16, mt = M / 16, nt = N / 16;
    const int ntile_m = mt / (WM * TM), ntile_n = nt / (WN * TN);
    if (mt % (WM * TM) || nt % (WN * TN)) { printf("%s: shape does not tile\n", name); return; }
    const int grid = ntile_m * ((ntile_n + 7) / 8) * 8;
    auto launch = [&] { hipLaunchKernelGGL((frag_gemm<WM, WN, TM, TN, RING>), dim3(grid), dim3(64 * WM * WN), 0, 0, dA, dB, dC, mt, nt, kg, N); };
    launch();
    CHECK(hipDeviceSynchronize());
    // spot check 64 entries against a float64 dot product
    std::vector<float> hC((size_t)M * N);
    CHECK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int s = 0; s < 64; ++s) {
        const int m = (s * 7919) % M, n = (s * 104729) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)m * K + k] * hB[(size_t)k * N + n];
        const double err = fabs(ref - hC[(size_t)m * N + n]);
        if (err > maxerr) maxerr = err;
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms / reps * 1e3;
    printf("{\"kernel\": \"%s\", \"tile\": \"%dx%d (%dx%d waves, %dx%d acc, ring %d)\", \"M\": %d, \"K\": %d, \"N\": %d, \"workgroups\": %d, \"us\": %.1f, \"tflops\": %.1f, \"max_abs_err\": %.2e}\n",
           name, WM * TM * 16, WN * TN * 16, WM, WN, TM, TN, RING, M, K, N, ntile_m * ntile_n, us, 2.0 * M * K * N / us / 1e6, maxerr);
    fflush(stdout);
}

int main()
{
    const int K = 1000, N = 12000;
    for (int M : {384, 6400}) {
        std::vector<float> hA((size_t)M * K), hB((size_t)K * N), pA, pB;
        srand(1);
        for (auto& x : hA) x = (float)rand() / RAND_MAX * 2.f - 1.f;
        for (auto& x : hB) x = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.1f;
        const int kg = (K + 15) / 16;
        pack_a(hA, M, K, kg, pA);
        pack_b(hB, K, N, kg, pB);
        float *dA, *dB, *dC;
        CHECK(hipMalloc(&dA, pA.size() * 4)); CHECK(hipMalloc(&dB, pB.size() * 4)); CHECK(hipMalloc(&dC, (size_t)M * N * 4));
        CHECK(hipMemcpy(dA, pA.data(), pA.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dB, pB.data(), pB.size() * 4, hipMemcpyHostToDevice));
        if (M == 384) {
            run<2, 2, 2, 3, 4>("pick-shape", M, K, N, dA, dB, dC, hA, hB);
            run<2, 2, 2, 3, 6>("pick-shape", M, K, N, dA, dB, dC, hA, hB);
            run<2, 2, 3, 3, 4>("pick-shape", M, K, N, dA, dB, dC, hA, hB);
            run<4, 1, 1, 6, 4>("pick-shape", M, K, N, dA, dB, dC, hA, hB);
            run<2, 2, 3, 5, 3>("pick-shape", M, K, N, dA, dB, dC, hA, hB);
        } else {
            run<2, 2, 4, 5, 3>("logits-shape", M, K, N, dA, dB, dC, hA, hB);
            run<2, 2, 5, 5, 2>("logits-shape", M, K, N, dA, dB, dC, hA, hB);
            run<2, 2, 4, 3, 4>("logits-shape", M, K, N, dA, dB, dC, hA, hB);
            run<2, 2, 2, 3, 6>("logits-shape", M, K, N, dA, dB, dC, hA, hB);
        }
        CHECK(hipFree(dA)); CHECK(hipFree(dB)); CHECK(hipFree(dC));
    }
    return 0;
}
