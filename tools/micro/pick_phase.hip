// pick_phase.hip -- feasibility probe (diagnostic only) for the vocabulary phase of a PERSISTENT decode loop: what one step's
// logits product costs when it is laid out the way a 250-workgroup persistent kernel would have to run it --
//   * one workgroup per CU, workgroup j owns 48 vocabulary columns (3 MFMA column tiles) for ALL M rows;
//   * the state h (M x K) arrives in MFMA A-fragment order ([row tile][k group][lane][4], what chain.hip exchanges), so every
//     wave streams the fragments of its own 6 row tiles straight into registers (no LDS, no sharing between waves);
//   * its 48 columns of W (row-major [K][N], re-streamed every step: 192 KB per workgroup does not fit LDS beside anything
//     else) go global -> LDS by buffer_load ... lds, 16 k-rows (3 KB) per stage, ring of NB stages, one barrier per stage.
// Prints time and TFLOP/s for M = 384, K = 1000, N = 12000 (the bench's pick: 97 us as 750 independent 64x96 tiles).
//   hipcc --offload-arch=gfx950 -O3 -o pick_phase tools/micro/pick_phase.hip && ./pick_phase
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

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

#ifndef PP_NB
#define PP_NB 4
#endif
#ifndef PP_RING
#define PP_RING 6
#endif
constexpr int TMW = 6, TNC = 3, NB = PP_NB;     // row tiles per wave, column tiles per workgroup, LDS stages
constexpr int RING = PP_RING;                   // k groups of A fragments in flight per wave (RING * TMW KB)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void pick_phase(const float* __restrict__ Afrag, const float* __restrict__ W,
                                                                                            float* __restrict__ C, int kg, int K, int N, int ldc)
{
    __shared__ __attribute__((aligned(16))) float Bs[NB][16 * 48];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int n0 = blockIdx.x * 48;
    f32x4 acc[TMW][TNC];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TNC; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // B stage: 16 rows x 48 floats = 192 float4 = three 1-KiB DMA pieces; wave w issues piece w (waves 0..2)
    const int q = wave * 64 + lane, brow = q / 12, bc4 = (q % 12) * 4;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto issue_b = [&](int g, int buf) __attribute__((always_inline)) {
#ifdef PP_NOB
        if (g >= NB - 1) return;
#endif
        if (wave < 3) {
            const int k = g * 16 + brow;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W + (size_t)g * 16 * N), 0, 0x7fffffff, 0x00020000);
            const unsigned off = (k < K && g < kg) ? (unsigned)(brow * N + n0 + bc4) * 4u : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(&Bs[buf][wave * 256]), 16, off, 0, 0, 0);
        }
    };
    // A fragments of this wave's row tiles: tile t, group g at ((t * kg + g) * 64 + lane) float4
    const f32x4* Ap = reinterpret_cast<const f32x4*>(Afrag) + (size_t)wave * TMW * kg * 64 + lane;
    f32x4 a[RING][TMW];
    static_for<0, RING>([&](auto r_) {
        constexpr int r = decltype(r_)::value;
#pragma unroll
        for (int i = 0; i < TMW; ++i) a[r][i] = Ap[((size_t)i * kg + (r < kg ? r : kg - 1)) * 64];
    });
    static_for<0, NB - 1>([&](auto g_) { issue_b(decltype(g_)::value, decltype(g_)::value); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // B fragments are read ONE GROUP AHEAD (stage g+1 while group g multiplies), so a stage must be visible two groups
    // before it is multiplied: the wait at the end of group g covers stage g+2.
    int buf = 0, nbuf = NB - 1;
    float bvc[4][TNC], bvn[4][TNC];
    {
        const float* b = &Bs[0][lq * 48 + l15];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < TNC; ++j) bvc[e][j] = b[e * 4 * 48 + j * 16];
    }
    for (int g0 = 0; g0 < kg; g0 += RING) {
        static_for<0, RING>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            const int g = g0 + r;
            if (g < kg) {
                issue_b(g + NB - 1, nbuf);
                const int b1 = buf + 1 == NB ? 0 : buf + 1;
                const float* b = &Bs[b1][lq * 48 + l15];
#ifndef PP_NOLDS
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < TNC; ++j) bvn[e][j] = b[e * 4 * 48 + j * 16];
#else
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < TNC; ++j) bvn[e][j] = bvc[e][j] + (float)(size_t)b * 0.f;
#endif
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, 4>([&](auto e_) {
                    constexpr int e = decltype(e_)::value;
#pragma unroll
                    for (int i = 0; i < TMW; ++i)
#pragma unroll
                        for (int j = 0; j < TNC; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][i][e], bvc[e][j], acc[i][j], 0, 0, 0);
                });
                __builtin_amdgcn_sched_barrier(0);
                const int gn = g + RING < kg ? g + RING : kg - 1;
#ifndef PP_NOA
#pragma unroll
                for (int i = 0; i < TMW; ++i) a[r][i] = Ap[((size_t)i * kg + gn) * 64];
#else
                (void)gn;
#endif
                __builtin_amdgcn_sched_barrier(0);
                // stage g+2 must have landed (its DMA was issued NB-3 groups ago); the A loads just issued stay in flight
                #if !defined(PP_NOA) && !defined(PP_NOB)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TMW + (NB - 3) * (TMW + 1)) : "memory");
#else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#ifndef PP_NOBAR
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#endif
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < TNC; ++j) bvc[e][j] = bvn[e][j];
                nbuf = buf;
                buf = b1;
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // epilogue: row-wise maximum of this workgroup's 48 columns per row (what a pick keeps) + optional full store for checking
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = (wave * TMW + i) * 16 + lq * 4 + r;
            if (C) {
#pragma unroll
                for (int j = 0; j < TNC; ++j) C[(size_t)m * ldc + n0 + j * 16 + l15] = acc[i][j][r];
            }
        }
}

int main()
{
    const int M = 384, K = 1000, N = 12000, kg = (K + 15) / 16;
    std::vector<float> hA((size_t)M * K), hW((size_t)K * N), pA((size_t)(M / 16) * kg * 256, 0.f);
    srand(1);
    for (auto& x : hA) x = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& x : hW) x = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.1f;
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k) pA[((size_t)(m / 16) * kg + k / 16) * 256 + ((k % 4) * 16 + m % 16) * 4 + (k % 16) / 4] = hA[(size_t)m * K + k];
    float *dA, *dW, *dC;
    CHECK(hipMalloc(&dA, pA.size() * 4)); CHECK(hipMalloc(&dW, hW.size() * 4 + 4096)); CHECK(hipMalloc(&dC, (size_t)M * N * 4));
    CHECK(hipMemcpy(dA, pA.data(), pA.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pick_phase, dim3(N / 48), dim3(256), 0, 0, dA, dW, dC, kg, K, N, N);
    CHECK(hipDeviceSynchronize());
    std::vector<float> hC((size_t)M * N);
    CHECK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int s = 0; s < 256; ++s) {
        const int m = (s * 7919) % M, n = (s * 104729) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)m * K + k] * hW[(size_t)k * N + n];
        maxerr = fmax(maxerr, fabs(ref - hC[(size_t)m * N + n]));
    }
    for (int store = 1; store >= 0; --store) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(pick_phase, dim3(N / 48), dim3(256), 0, 0, dA, dW, store ? dC : nullptr, kg, K, N, N);
        CHECK(hipEventRecord(e0));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(pick_phase, dim3(N / 48), dim3(256), 0, 0, dA, dW, store ? dC : nullptr, kg, K, N, N);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms / reps * 1e3;
        printf("{\"kernel\": \"pick-phase probe\", \"layout\": \"250 workgroups x (384 rows x 48 columns), A in fragment order from L2, W by LDS-DMA (%d stages), ring %d\", \"store\": %s, \"us\": %.1f, \"tflops\": %.1f, \"max_abs_err\": %.2e}\n",
               NB, RING, store ? "true" : "false", us, 2.0 * M * K * N / us / 1e6, maxerr);
    }
    return 0;
}
