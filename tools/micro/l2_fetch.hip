// Dev microbenchmark: how many bytes per clock can one CU pull from L2 with the access pattern of the step kernels
// (a workgroup streams an A panel [rows x K] and a W panel [K x cols], 16-byte loads, D loads in flight per thread)?
// hipcc --offload-arch=gfx950 -O3 -o l2_fetch l2_fetch.hip && ./l2_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(256) void fetch(const float* A, const float* W, int K, int lda, int ldw, int rows, int cols, int ntn, float* out, int reps)
{
    const int tid = threadIdx.x;
    const int tile_n = blockIdx.x % ntn, tile_m = (blockIdx.x / ntn) % 4;
    const float* a = A + (size_t)tile_m * rows * lda;
    const float* w = W + tile_n * cols;
    f32x4 acc = {0, 0, 0, 0};
    // per chunk of 32 k: A rows x 32 (rows*8 float4), W 32 x cols (32*cols/4 float4)
    const int na = rows * 8, nw = 8 * cols, per = na + nw;
    for (int r = 0; r < reps; ++r)
        for (int k0 = 0; k0 < K; k0 += 32 * D) {
            f32x4 v[D * 5];
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const int idx = tid + i * 256;
                    const int k = k0 + d * 32;
                    const float* p = idx < na ? a + (size_t)(idx / 8) * lda + k + (idx % 8) * 4
                                   : (idx < per ? w + (size_t)(k + (idx - na) / (cols / 4)) * ldw + ((idx - na) % (cols / 4)) * 4 : a);
                    v[d * 5 + i] = *reinterpret_cast<const f32x4*>(p);
                }
#pragma unroll
            for (int i = 0; i < D * 5; ++i) acc += v[i];
        }
    if (acc[0] == 123.456f) out[0] = acc[1] + acc[2] + acc[3];
}

int main()
{
    const int M = 320, K = 1024, N = 4032, rows = 80, cols = 64;
    float *A, *W, *out;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)K * N * 4); hipMalloc(&out, 64);
    hipMemset(A, 0, (size_t)M * K * 4); hipMemset(W, 0, (size_t)K * N * 4);
    const int ntn = (N + cols - 1) / cols, ntm = M / rows;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, int D, int grid) {
        const int reps = 4;
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, A, W, K, K, N, rows, cols, ntn, out, reps);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, A, W, K, K, N, rows, cols, ntn, out, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)grid * reps * (rows + cols) * 4.0 * K;
        printf("D=%d grid=%d: %.1f us, %.2f TB/s aggregate, %.1f B/clk/CU @2.1GHz (256 CUs)\n", D, grid, ms * 1e3, bytes / ms / 1e9,
               bytes / (ms * 1e-3) / 256 / 2.1e9);
    };
    run(fetch<1>, 1, ntm * ntn); run(fetch<2>, 2, ntm * ntn); run(fetch<4>, 4, ntm * ntn); run(fetch<8>, 8, ntm * ntn);
    run(fetch<4>, 4, 2 * ntm * ntn); run(fetch<8>, 8, 2 * ntm * ntn);
    return 0;
}
