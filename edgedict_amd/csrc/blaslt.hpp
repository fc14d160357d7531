// Vendor-library route for the PLAIN large bf16 product of the joint (logits = hid W2^T + b2).
// Internal interface; see blaslt.cpp.
#pragma once
#include <hip/hip_runtime.h>
// true if hipBLASLt took the product (enqueued on `s`); false -> caller runs its own kernel
bool ed_blaslt_nt_bf16(const void* A, long long lda, const void* B, long long ldb, void* C,
                       long long ldc, int M, int N, int K, const float* bias, int accumulate,
                       hipStream_t s);
// C[M,N] fp32 (+)= A[K,M]^T B[K,N], bf16 operands stored K-major-strided (weight gradients)
bool ed_blaslt_tn_f32(const void* A, long long lda, const void* B, long long ldb, float* C,
                      long long ldc, int M, int N, int K, int accumulate, hipStream_t s);
