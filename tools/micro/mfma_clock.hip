// mfma_clock.hip -- what the fp32 matrix pipe of THIS chip sustains, and at what clock (MI355X_MICROARCH.md, DVFS item 6).
//
// Diagnostic only (nothing in the product includes it).  Three loops of v_mfma_f32_16x16x4_f32 on random operands, one
// workgroup of 4 (or 8) waves per CU slot, every CU busy:
//   bare : operands in registers, TM x TN independent accumulators per wave -- the instruction's ceiling
//   lds  : every k-step's fragments re-read from LDS the way gemm_mfma.h reads them (A: ds_read_b128 per four k-steps
//          from k%4 planes, B: ds_read_b32 per k-step), no global traffic, no barrier -- the ceiling of the operand path
//   ldsb : the same with one workgroup barrier per 32-deep chunk (the product's chunk protocol without its loads)
// Each wave stamps s_memtime / s_memrealtime around its loop: in-kernel clock = d(memtime) / d(memrealtime) x 100 MHz
// (median over waves), after >= 2 s of back-to-back launches.  Output: one JSON object per configuration.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_clock tools/micro/mfma_clock.hip && ./mfma_clock
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } \
    } while (0)

struct Stamp { unsigned long long clk, real; };

template <int TM, int TN, int MODE>   // MODE 0 bare, 1 lds, 2 lds + barrier per chunk
__global__ __launch_bounds__(512) void mfma_loop(const float* __restrict__ rnd, float* __restrict__ sink, Stamp* stamps, int chunks)
{
    constexpr int KQ = 8, BM = 128, BN = 128;                      // a 32-deep chunk of a 128 x 128 tile image
    extern __shared__ __attribute__((aligned(16))) float smem[];   // A planes [4][BM][KQ] + B [32][BN + 16]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a[TM], b[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a[i] = rnd[(threadIdx.x * 8 + i) & 4095];
#pragma unroll
    for (int j = 0; j < TN; ++j) b[j] = rnd[(threadIdx.x * 8 + 4 + j + blockIdx.x) & 4095];
    if (MODE > 0) {
        for (int i = threadIdx.x; i < 4 * BM * KQ + 32 * (BN + 16); i += blockDim.x) smem[i] = rnd[(i * 7 + blockIdx.x) & 4095];
        __syncthreads();
    }
    const float* As = smem + lq * BM * KQ + ((wave & 1) * TM * 16 + l15) * KQ;
    const float* Bs = smem + 4 * BM * KQ + lq * (BN + 16) + ((wave >> 1) & 1) * TN * 16 + l15;
    const int swz = (l15 >> 3) & 1;
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int c = 0; c < chunks; ++c) {
        if (MODE == 0) {
#pragma unroll
            for (int ks = 0; ks < KQ; ++ks)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        } else {
            f32x4 a4[2][TM];
            float bv[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a4[0][i] = *reinterpret_cast<const f32x4*>(As + i * 16 * KQ + ((0 ^ swz) << 2));
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[0][j] = Bs[j * 16];
#pragma unroll
            for (int ks = 0; ks < KQ; ++ks) {
                if (ks == 0) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) a4[1][i] = *reinterpret_cast<const f32x4*>(As + i * 16 * KQ + ((1 ^ swz) << 2));
                }
                if (ks + 1 < KQ) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) bv[(ks + 1) & 1][j] = Bs[(ks + 1) * 4 * (BN + 16) + j * 16];
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[ks / 4][i][ks % 4], bv[ks & 1][j], acc[i][j], 0, 0, 0);
            }
            if (MODE == 2) __syncthreads();
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 123.456f) sink[0] = s;                                 // keeps the accumulators alive; never true on this data
    if (lane == 0) stamps[blockIdx.x * (blockDim.x >> 6) + wave] = Stamp{t1 - t0, r1 - r0};   // a buffer of their own
}

template <int TM, int TN, int MODE>
void run(const char* name, int waves_per_wg, int wgs_per_cu, const float* rnd, float* sink, Stamp* stamps, int chunks)
{
    const int ncu = 256, grid = ncu * wgs_per_cu, block = waves_per_wg * 64;
    const size_t lds = MODE > 0 ? (4 * 128 * 8 + 32 * (128 + 16)) * sizeof(float) : 0;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto launch = [&] { hipLaunchKernelGGL((mfma_loop<TM, TN, MODE>), dim3(grid), dim3(block), lds, 0, rnd, sink, stamps, chunks); };
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms1 = 0.f;
    CHECK(hipEventElapsedTime(&ms1, e0, e1));
    const int heat = (int)(2200.0f / ms1) + 1;                      // >= 2 s back to back before the measured launches
    for (int i = 0; i < heat; ++i) launch();
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const int nw = grid * waves_per_wg;
    std::vector<Stamp> h(nw);
    CHECK(hipMemcpy(h.data(), stamps, nw * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> ghz(nw), cyc(nw);
    for (int i = 0; i < nw; ++i) { ghz[i] = (double)h[i].clk / (double)h[i].real * 0.1; cyc[i] = (double)h[i].clk; }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    const double mfmas = (double)chunks * 8 * TM * TN;              // per wave
    const double flops = mfmas * 2048.0 * nw;
    printf("{\"loop\": \"%s\", \"waves_per_simd\": %.1f, \"TMxTN\": \"%dx%d\", \"ms\": %.3f, \"tflops\": %.1f, \"in_kernel_clock_ghz_median\": %.3f, "
           "\"clock_p10\": %.3f, \"clock_p90\": %.3f, \"cycles_per_mfma_median\": %.2f, \"tflops_at_2.4GHz_same_cycles\": %.1f}\n",
           name, waves_per_wg * wgs_per_cu / 4.0, TM, TN, ms, flops / (ms * 1e-3) / 1e12, ghz[nw / 2], ghz[nw / 10], ghz[nw * 9 / 10],
           cyc[nw / 2] / mfmas * (waves_per_wg * wgs_per_cu > 4 ? 4.0 / (waves_per_wg * wgs_per_cu) : 1.0),
           flops / (ms * 1e-3) / 1e12 * 2.4 / ghz[nw / 2]);
    fflush(stdout);
}

int main()
{
    float *rnd, *sink;
    Stamp* stamps;
    CHECK(hipMalloc(&rnd, 4096 * sizeof(float)));
    CHECK(hipMalloc(&sink, 16));
    CHECK(hipMalloc(&stamps, 256 * 8 * 8 * sizeof(Stamp)));
    std::vector<float> h(4096);
    srand(1);
    for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f;
    CHECK(hipMemcpy(rnd, h.data(), 4096 * sizeof(float), hipMemcpyHostToDevice));
    const int chunks = 20000;                                        // 2.56 M MFMAs per wave at 4x4: ~35 ms per launch
    run<4, 4, 0>("bare", 4, 1, rnd, sink, stamps, chunks);
    run<4, 4, 0>("bare", 4, 2, rnd, sink, stamps, chunks / 2);
    run<4, 4, 1>("lds", 4, 1, rnd, sink, stamps, chunks);
    run<4, 4, 1>("lds", 4, 2, rnd, sink, stamps, chunks / 2);
    run<4, 4, 2>("lds+barrier", 4, 1, rnd, sink, stamps, chunks);
    run<4, 4, 2>("lds+barrier", 4, 2, rnd, sink, stamps, chunks / 2);
    run<3, 3, 2>("lds+barrier", 4, 2, rnd, sink, stamps, chunks / 2);
    run<5, 1, 2>("lds+barrier", 4, 1, rnd, sink, stamps, chunks);
    return 0;
}
