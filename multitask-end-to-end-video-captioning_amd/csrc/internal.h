// internal.h -- launcher interfaces shared by the translation units of libs2vt_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_mfma.h"

namespace s2vt {

// forward contraction (exact ascending-k chain); epi in {EPI_STORE, EPI_LSTM, EPI_PICK}.
// cfg < 0: pick a tile configuration from the shape; >= 0 forces table entry `cfg` (tests, tuning).
hipError_t launch_gemm(const GemmArgs& a, int epi, int cfg, hipStream_t st);
int gemm_num_cfgs(int epi);
const char* gemm_cfg_name(int epi, int cfg);

// C[Kout, N] (+)= sum_m A[row(m), k] * B[m, n]   (weight gradients; order-free, fp32 MFMA)
struct TnArgs {
    const float* A; const int* rowidx; int lda;       // [Mred, Kout] (rows optionally gathered)
    const float* B; int ldb;                           // [Mred, N]
    float* C; int ldc;                                 // [Kout, N]
    int Mred, Kout, N;
    int accumulate;                                    // 1: C += result (fp32 atomics when split)
};
hipError_t launch_gemm_tn(const TnArgs& a, hipStream_t st);

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
