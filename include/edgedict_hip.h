/*
 * edgedict_hip.h — C ABI of libedgedict_hip.so, the MI355X (gfx950) RNN-Transducer engine.
 *
 * This is the drop-in boundary one level below the Python class surface
 * (rnnt.models.Transducer / rnnt.stream.PytorchStreamDecoder).  Every entry point
 *   - takes raw DEVICE pointers, sizes and a hipStream_t (passed as void*),
 *   - allocates nothing: the caller (the PyTorch caching allocator in our host code)
 *     owns every buffer including workspaces,
 *   - keeps no global mutable state, so it is re-entrant per device/stream
 *     (nn.DataParallel drives one Python thread per GPU in one process),
 *   - never aborts: it returns ED_OK or a negative status and leaves a message in
 *     edgedict_last_error() (thread-local) which the binding raises as RuntimeError.
 *
 * dtype codes: activations/weights are either fp32 (parity mode) or bf16 (throughput mode);
 * reductions, LSTM cell state, softmax denominators, alpha/beta and costs are always fp32.
 *
 * Each block below cites the reference interface (file:line under the upstream repo)
 * whose arithmetic it replaces.
 */
#ifndef EDGEDICT_HIP_H
#define EDGEDICT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EDGEDICT_ABI_VERSION 1

#define ED_OK 0
#define ED_ERR_INVALID (-1) /* bad argument (shape / dtype / alignment) */
#define ED_ERR_LAUNCH (-2)  /* HIP runtime reported an error */

#define ED_F32 0
#define ED_BF16 1

const char* edgedict_last_error(void);
int edgedict_abi_version(void);

/* ------------------------------------------------------------------------------------
 * RNN-T loss.  Replaces warprnnt_pytorch.RNNTLoss(blank) as called at
 * rnnt/models.py:221,238 and cli/lightning.py:40,91 (third-party, un-vendored; its public
 * C entry points are compute_rnnt_loss / get_workspace_size, which this pair mirrors).
 *
 *   acts        [B, T, U1, V]  raw joint logits (NOT log-softmaxed), contiguous, acts_dtype
 *   labels      [B, U1-1]      int32, padded arbitrarily beyond label_lens[b]
 *   act_lens    [B] int32      valid frames per utterance   (1 <= act_lens[b] <= T)
 *   label_lens  [B] int32      valid labels per utterance   (0 <= label_lens[b] <= U1-1)
 *   costs       [B] fp32       out: -log P(y|x) per utterance
 *   reduced     [1] fp32       out (nullable): reduce_scale * sum_b costs[b]  ('mean' => 1/B)
 *   workspace   edgedict_rnnt_workspace_bytes(B,T,U1) bytes, 16-byte aligned; forward fills
 *               it (log-softmax denominators, alpha, beta, log-likelihoods) and backward
 *               reads it, so it must be kept alive between the two calls.
 *
 * backward writes d(sum_b scale*cost_b)/d(acts) into grads (same shape/dtype as acts):
 *   scale_b = grad_scale_host * (grad_scale_dev ? grad_scale_dev[b*grad_scale_stride] : 1)
 * (stride 0 = one scalar for the batch, stride 1 = per-utterance, reduction 'none') so that the 'mean' reduction's 1/B and autograd's incoming grad_output need no host sync.
 * Cells outside an utterance's (act_lens, label_lens+1) box get exact zeros.
 * Limits: U1 <= 1024; V*sizeof(dtype) must be a multiple of 16 for the vector path
 * (other V take a scalar path).
 */
size_t edgedict_rnnt_workspace_bytes(int B, int T, int U1);
int edgedict_rnnt_loss_forward(const void* acts, int acts_dtype, const int32_t* labels,
                               const int32_t* act_lens, const int32_t* label_lens, int B, int T,
                               int U1, int V, int blank, float* costs, float* reduced,
                               float reduce_scale, void* workspace, void* stream);
int edgedict_rnnt_loss_backward(const void* acts, int acts_dtype, void* grads,
                                const int32_t* labels, const int32_t* act_lens,
                                const int32_t* label_lens, int B, int T, int U1, int V, int blank,
                                const void* workspace, float grad_scale_host,
                                const float* grad_scale_dev, int grad_scale_stride, void* stream);
/* debug / test accessors into a filled workspace (device pointers):
 * which: 0 = log-softmax denominators f32[B,T,U1], 1 = alphas f64[B,T,U1], 2 = betas f64,
 * 3 = log-likelihoods f64[B,2] (alpha-side, beta-side), 4 = lp_blank f32[B,T,U1], 5 = lp_label */
const void* edgedict_rnnt_workspace_view(const void* workspace, int B, int T, int U1, int which);

/* ------------------------------------------------------------------------------------
 * Dense product on the matrix cores:  C[M,N] (+)= A[M,K] * B[N,K]^T + bias1[N] + bias2[N].
 * Replaces the cuBLAS/cuDNN GEMMs behind nn.Linear / nn.LSTM input products
 * (rnnt/models.py:45-46,65,129,135,148,156,165-167) and their autograd transposes.
 *   x_kmajor = 1: element (row,k) at p[row*ld + k];  x_kmajor = 0: at p[k*ld + row]
 *   dtype_in : ED_BF16 (v_mfma_f32_16x16x32_bf16) or ED_F32 (exact v_mfma_f32_16x16x4_f32)
 *   dtype_out: ED_F32 or (bf16 inputs only) ED_BF16; accumulation is always fp32
 *   bias1/bias2: nullable fp32 [N];  accumulate != 0: C += ...;
 *   split_k > 1: K is partitioned over workgroups and combined with fp32 atomics
 *                (fp32 output only; used for weight gradients whose M*N is small and K huge)
 */
int edgedict_gemm(int dtype_in, int dtype_out, const void* A, long long lda, int a_kmajor,
                  const void* B, long long ldb, int b_kmajor, void* C, long long ldc, int M, int N,
                  int K, const float* bias1, const float* bias2, int accumulate, int split_k,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EDGEDICT_HIP_H */
