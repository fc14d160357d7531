// bf16 NT fast path (gemm_nt.hip): direct-to-LDS double-buffered tiles.  Internal interface.
#pragma once
#include <hip/hip_runtime.h>
bool ed_gemm_nt_ok(int dtype_in, int dtype_out, const void* A, long long lda, int a_kmajor,
                   const void* B, long long ldb, int b_kmajor, const void* C, long long ldc, int M,
                   int N, int K, int split_k, const float* bias1, const float* bias2);
int ed_gemm_nt_launch(const void* A, long long lda, const void* B, long long ldb, void* C,
                      long long ldc, int M, int N, int K, const float* bias1, const float* bias2,
                      int accumulate, int lds_pad, hipStream_t s);
// 256 x 256 macro-tile variant (gemm_nt256.hip) for large products
bool ed_gemm_nt256_ok(int M, int N, int K, int accumulate);
bool ed_gemm_nt256_shape_ok(int M, int N, int K);
// lse_part (nullable): [M][ceil(N/64)][2] fp32 = (max, sum exp(x - max)) of the bf16-rounded C values of
// each row over 64-column slots
int ed_gemm_nt256_launch(const void* A, long long lda, const void* B, long long ldb, void* C,
                         long long ldc, int M, int N, int K, const float* bias1, const float* bias2,
                         hipStream_t s, float* lse_part = nullptr);
// persistent ring variant of the same product (gemm_nt256r.hip, round 6; EDGEDICT_GEMM_NT256R=0 switches it off)
bool ed_gemm_nt256r_ok(int M, int N, int K, bool has_bias, int* grid_out = nullptr);
int ed_gemm_nt256r_launch(const void* A, long long lda, const void* B, long long ldb, void* C, long long ldc, int M,
                          int N, int K, const float* bias1, const float* bias2, hipStream_t s, float* lse_part);
// 256 x 128-tile TN kernel for weight gradients (gemm_tn256.hip): P[s][M,N] = sum_{k in slice s} A[k,m] B[k,n],
// fp32 K-slice partials written once each ("quiet"); the caller sums the `slices` slices
bool ed_gemm_tn256_ok(const void* A, long long lda, const void* B, long long ldb, int M, int N, int K);
int ed_gemm_tn256_slices(int M, int N, int K, int max_slices);
int ed_gemm_tn256_partials(const void* A, long long lda, const void* B, long long ldb, float* partials, int M,
                           int N, int K, int slices, int max_wgs, hipStream_t s);
