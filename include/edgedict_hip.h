/*
 * edgedict_hip.h — C ABI of libedgedict_hip.so, the MI355X (gfx950) RNN-Transducer engine.
 *
 * This is the drop-in boundary one level below the Python class surface
 * (rnnt.models.Transducer / rnnt.stream.PytorchStreamDecoder).  Every entry point
 *   - takes raw DEVICE pointers, sizes and a hipStream_t (passed as void*),
 *   - allocates no caller-visible buffer: the caller (the PyTorch caching allocator in our host
 *     code) owns every tensor and workspace it passes in,
 *   - keeps exactly this per-DEVICE state, created lazily under a mutex and never exposed: three
 *     internal HIP streams + an event pool for the encoder-stack scheduler (edgedict_aux_stream,
 *     edgedict_stack_*), a pinned host word for the give-up codes of bounded in-kernel waits
 *     (edgedict_stack_wsr_error), the timing record read by edgedict_stack_last_timing /
 *     edgedict_stack_launch_times (two 64 KB device buffers, allocated only once
 *     edgedict_stack_time_launches(1) was called), and - only with EDGEDICT_BLASLT=1 in the
 *     environment, when a product is routed to the vendor library - one hipBLASLt handle per device
 *     with a 64 MiB workspace per (device, stream).  Entry points are re-entrant per device; calls that touch
 *     the same device from several host threads must be serialised by the caller
 *     (nn.DataParallel's one-thread-per-GPU pattern is fine: different devices),
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
/* packed-lattice forward (bf16 logits) with the per-row log-sum-exp partials produced by
 * edgedict_gemm_nt_lse: lse_parts [M_valid][lse_slots][2], lse_slots = ceil(V/64).  Same results as
 * edgedict_rnnt_loss_forward_packed up to the summation order of the denominators. */
int edgedict_rnnt_loss_forward_packed_parts(const void* acts, const int32_t* labels,
                                            const int32_t* act_lens, const int32_t* label_lens,
                                            const long long* row_offsets, int B, int T, int U1,
                                            int V, int blank, float* costs, float* reduced,
                                            float reduce_scale, void* workspace,
                                            const float* lse_parts, int lse_slots, void* stream);
int edgedict_rnnt_loss_backward(const void* acts, int acts_dtype, void* grads,
                                const int32_t* labels, const int32_t* act_lens,
                                const int32_t* label_lens, int B, int T, int U1, int V, int blank,
                                const void* workspace, float grad_scale_host,
                                const float* grad_scale_dev, int grad_scale_stride, void* stream);
/* PACKED lattice variants: `acts` / `grads` hold ONLY the cells inside each utterance's box, row of
 * cell (b,t,u) = row_offsets[b] + t*(label_lens[b]+1) + u (int64 device array [B], exclusive prefix
 * sums of act_lens[b]*(label_lens[b]+1)); T, U1 still bound the (dense, small) lattice workspace.
 * The joint network then only multiplies valid rows: on the E6D2 bench batch 65 % of the dense
 * B*T*U1 rows.  Arithmetic per cell is identical to the dense entry points. */
int edgedict_rnnt_loss_forward_packed(const void* acts, int acts_dtype, const int32_t* labels,
                                      const int32_t* act_lens, const int32_t* label_lens,
                                      const long long* row_offsets, int B, int T, int U1, int V,
                                      int blank, float* costs, float* reduced, float reduce_scale,
                                      void* workspace, void* stream);
int edgedict_rnnt_loss_backward_packed(const void* acts, int acts_dtype, void* grads,
                                       const int32_t* labels, const int32_t* act_lens,
                                       const int32_t* label_lens, const long long* row_offsets,
                                       int B, int T, int U1, int V, int blank, const void* workspace,
                                       float grad_scale_host, const float* grad_scale_dev,
                                       int grad_scale_stride, void* stream);
/* edgedict_rnnt_loss_backward_packed that also leaves the COLUMN SUMS of the gradient matrix it writes (of the fp32 values
 * before they are rounded to the gradient's dtype) as `edgedict_rnnt_grad_colsum_rows(...)` partial rows of V floats in
 * colsum_parts: their sum over rows is the joint's output-bias gradient (rnnt/models.py:165-167 through autograd), which
 * otherwise takes a second pass over the matrix (2.2 GB at the E6D2 bench batch).  _rows returns 0 when the fused form
 * is not available (V * sizeof(element) not a multiple of 16, or V > 2048 for bf16 / 1024 for f32). */
int edgedict_rnnt_grad_colsum_rows(int acts_dtype, int B, int T, int U1, int V);
int edgedict_rnnt_loss_backward_packed_colsum(const void* acts, int acts_dtype, void* grads,
                                              const int32_t* labels, const int32_t* act_lens,
                                              const int32_t* label_lens, const long long* row_offsets,
                                              int B, int T, int U1, int V, int blank, const void* workspace,
                                              float grad_scale_host, const float* grad_scale_dev,
                                              int grad_scale_stride, float* colsum_parts, void* stream);
/* the same for utterances [b0, b0 + nb) of the batch only (all array arguments are still the whole batch's):
 * lets the host pipeline the gradient of one group of utterances (HBM-bound) against the joint's dhid product
 * of the previous group (matrix-pipe-bound) on a second stream. */
int edgedict_rnnt_loss_backward_packed_range(const void* acts, int acts_dtype, void* grads,
                                             const int32_t* labels, const int32_t* act_lens,
                                             const int32_t* label_lens, const long long* row_offsets,
                                             int B, int T, int U1, int V, int blank, const void* workspace,
                                             float grad_scale_host, const float* grad_scale_dev,
                                             int grad_scale_stride, int b0, int nb, void* stream);
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

/* Background variant for products that are OFF the critical path (weight gradients) and run on a
 * side stream next to latency-bound kernels.  Identical arithmetic, three scheduling differences:
 *   - every workgroup claims enough extra LDS that at most max_wg_per_cu are resident per CU, and
 *     only as many workgroups are launched as are resident at once (each walks its work items): a
 *     long-lived GEMM can neither fill a CU nor park a tail of workgroups in the dispatcher;
 *   - with `partials` (fp32 [P][M][N], P = smallest power of two >= split_k, at most 8) the K
 *     slices are written ONCE, at the end of each workgroup, with plain stores and summed by a
 *     small reduce kernel - no atomics, no dirty lines while the main loop runs.  Measured on
 *     MI355X (tools/boundary_probe.hip): a concurrent kernel that streams writes / fp32 atomics
 *     raises every dependent kernel boundary on the other streams from 2.4 us to 10 / 8-32 us
 *     (the end-of-kernel L2 write-back has to flush everybody's dirty lines); reads cost nothing. */
int edgedict_gemm_bg(int dtype_in, int dtype_out, const void* A, long long lda, int a_kmajor,
                     const void* B, long long ldb, int b_kmajor, void* C, long long ldc, int M,
                     int N, int K, const float* bias1, const float* bias2, int accumulate,
                     int split_k, int max_wg_per_cu, float* partials, void* stream);

/* C[M,N] = A[M,K] B[N,K]^T + bias (bf16, both operands K-contiguous: the joint's logits product,
 * rnnt/models.py:177) with the log-softmax partials of every row fused into the epilogue:
 * lse_part [M][ceil(N/64)][2] fp32 = (max, sum exp(x - max)) of the bf16-rounded outputs over 64-column
 * slots - edgedict_rnnt_loss_forward_packed_parts finishes the denominators from them instead of
 * re-reading the M x N logits.  K %% 64 == 0, K >= 128, N %% 8 == 0. */
int edgedict_gemm_nt_lse(const void* A, long long lda, const void* B, long long ldb, void* C,
                         long long ldc, int M, int N, int K, const float* bias, float* lse_part,
                         void* stream);


/* ------------------------------------------------------------------------------------
 * LayerNorm fused with the residual add and the encoder's TimeReduction.
 * Replaces nn.LayerNorm at rnnt/models.py:124,132 and the per-layer
 * `xs = xs + lstm(xs); xs = Sequential(LayerNorm[, TimeReduction])(xs)` at rnnt/models.py:47-53,
 * 66-70 with TimeReduction.forward rnnt/models.py:21-29 (zero-pad AFTER the norm, pair mean).
 *   x, res (nullable)  [B,T,D] dtype;  s = x + res is what gets normalised
 *   y                  [B, ceil(T/reduce), D] dtype;  reduce in {1,2}
 *   mean, rstd         fp32 [B*T] saved for backward
 * backward: ds [B,T,D] dtype (gradient wrt x AND wrt res), dgamma/dbeta fp32 [D] are ACCUMULATED
 * (atomic +=), pass NULL to skip.
 */
int edgedict_layernorm_fwd(int dtype, const void* x, const void* res, const float* gamma,
                           const float* beta, void* y, float* mean, float* rstd, int B, int T,
                           int D, int reduce, float eps, void* stream);
int edgedict_layernorm_bwd(int dtype, const void* dout, const void* x, const void* res,
                           const float* gamma, const float* mean, const float* rstd, void* ds,
                           float* dgamma, float* dbeta, int B, int T, int D, int reduce,
                           void* stream);

/* ------------------------------------------------------------------------------------
 * LSTM time recurrence.  Replaces the cuDNN RNN behind nn.LSTM (rnnt/models.py:45-46,65 encoder
 * layers; rnnt/models.py:145-147,155 prediction network).  PyTorch gate order i,f,g,o.
 * forward:
 *   G      [B,T,4H] dtype  in : X*W_ih^T + b_ih + b_hh (from edgedict_gemm)
 *                          out: post-activation gates i,f,g,o (saved for backward, in place)
 *   Hprev  [B,T,H]  dtype  out: h_{t-1} per step (row t=0 <- h0); operand of dW_hh later
 *   Y      [B,T,H]  dtype  out: h_t
 *   Cst    [B,T,H]  fp32   out: c_t
 *   Whh    [4H,H]   dtype;  h0,c0 fp32 [B,H] nullable (= zeros);  hN,cN fp32 [B,H] nullable
 * backward (after forward, same buffers):
 *   dY     [B,T,H]  dtype  gradient wrt h_t (nullable = zeros)
 *   G                     in : saved gates;  out: dL/d(pre-activations) for every t (in place)
 *   WhhT   [H,4H]   dtype  (edgedict_transpose of Whh);  dC_ws fp32 [B,H] scratch
 * One kernel launch per timestep (kernel boundary ~1.5us < any grid barrier on MI355X).
 * Limits: H % 8 == 0.  Gradients wrt (h0,c0) are not produced.
 *
 * bf16 fast path (H % 32 == 0): pass the fragment-order weight image built by
 * edgedict_lstm_pack_weights (packed_fwd for forward, packed_bwd for backward; each 4H*H bf16,
 * rebuilt whenever W_hh changes) and a workspace of edgedict_lstm_workspace_bytes(dtype,B,H)
 * bytes (ping-pong fragment images of h_t / dG_t); the plain Whh / WhhT may then be NULL.
 * With NULL packed weights or workspace the generic path runs (any dtype).
 */
size_t edgedict_lstm_workspace_bytes(int dtype, int B, int H);
int edgedict_lstm_pack_weights(int src_dtype, const void* Whh, void* packed_fwd, void* packed_bwd,
                               int H, void* stream);
int edgedict_lstm_forward(int dtype, void* G, void* Hprev, void* Y, float* Cst, const void* Whh,
                          const void* Whh_packed, const float* h0, const float* c0, float* hN,
                          float* cN, int B, int T, int H, void* ws, void* stream);
int edgedict_lstm_backward(int dtype, void* G, const void* dY, const float* Cst, const float* c0,
                           const void* WhhT, const void* WhhT_packed, float* dC_ws, int B, int T,
                           int H, void* ws, void* stream);

/* ------------------------------------------------------------------------------------
 * Layer-pipelined LSTM encoder stack (bf16).  One call per direction replaces the whole of
 * Encoder.forward's LayerNorm + ResLayerNormLSTM.forward (rnnt/models.py:55-75,124,131-134:
 * per layer nn.LSTM, residual add for layers > 0, LayerNorm, optional TimeReduction
 * rnnt/models.py:21-29) and its autograd.  The layers run as a skewed wavefront: one launch
 * carries a time step of every runnable layer, the input products are chunked GEMMs on an
 * internal side stream, weight gradients overlap the BPTT of the layers below
 * (csrc/encoder_stack.hip).  The caller's stream is forked at entry and joined at exit.
 *
 * Layouts (all device memory, owned by the caller, kept from forward to backward):
 *   activations are TIME-MAJOR;  gate columns of G are interleaved: column of (gate g, unit j)
 *   = (j/16)*64 + g*16 + j%16  (gate order i,f,g,o);  weight images come from
 *   edgedict_stack_pack_weights (rebuilt whenever the fp32 master weights change).
 * Limits: H % 32 == 0, H <= 2048, L <= 8, reduce in {1,2}, residual layers need I == H.
 * The library keeps its internal streams (recurrence, chunk GEMMs, weight gradients)
 * and an event pool per device (created on first use).
 */
typedef struct edgedict_stack_layer {
    int T;                 /* time steps of this layer */
    int I;                 /* input width */
    int reduce;            /* time reduction after this layer's LayerNorm: 1 or 2 */
    int residual;          /* LayerNorm(y + x) instead of LayerNorm(y) */
    const void* wih_p;     /* bf16 [4H, I]  rows in interleaved gate order */
    const void* wih_t;     /* bf16 [I, 4H]  its transpose (K-contiguous operand of the dX product); backward only */
    const float* bias_p;   /* f32  [4H]     b_ih + b_hh, interleaved */
    const void* whh_f;     /* bf16 forward fragment image of W_hh (4H*H) */
    const void* whh_b;     /* bf16 backward fragment image of W_hh (4H*H); backward only */
    const float* ln_gamma; /* f32 [H] */
    const float* ln_beta;  /* f32 [H] */
    void* X;               /* bf16 [T, B, I]    layer input (layer 0: written by the input LayerNorm) */
    void* G;               /* bf16 [T, B, 4H]   gates after forward, dL/d(pre-activation) after backward */
    void* Yx;              /* bf16 [T+1, B, H]  row 0 = h0, row t+1 = h_t */
    float* Cx;             /* f32  [T+1, B, H]  row 0 = c0, row t+1 = c_t */
    float* mean;           /* f32  [T, B] LayerNorm statistics */
    float* rstd;
    /* backward only */
    void* dZ;              /* bf16 [T, B, H] dL/d(LayerNorm input) = dL/dh_t from above */
    void* dX;              /* bf16 [T, B, I] dL/d(layer input); pass dZ itself for residual layers
                              (the product is accumulated in place); unused for layer 0 */
    float* dW_ih;          /* f32 [4H, I] out, natural gate order */
    float* dW_hh;          /* f32 [4H, H] out */
    float* db;             /* f32 [4H] out (gradient of b_ih and of b_hh) */
    float* db_hh;          /* nullable second destination of the same values (b_hh.grad when accumulating) */
    float* dgamma;         /* f32 [H] ACCUMULATED (+=): zero before the call */
    float* dbeta;
    const void* whh_s;     /* bf16 split-K fragment image of W_hh (edgedict_stack_pack_sk), nullable: with it (and B <= 64,
                              H % 64 == 0, H <= 1024, EDGEDICT_STACK_BWD_SK != 0) the BPTT runs on the split-K
                              weights-stationary kernel, several steps per launch; without it one launch per step */
    /* nn.Dropout behind this layer's LayerNorm (+ TimeReduction), rnnt/models.py:47-53,70 - training mode only, the
       caller passes 0 in eval mode: element (b, tau, j) of the layer's [B, ceil(T / reduce), H] output is zeroed with
       probability drop_p and scaled by 1 / (1 - drop_p) otherwise; the mask is the counter-based one of
       edgedict_dropout (same hash, same batch-first element index, drop_seed), regenerated in the backward pass */
    float drop_p;
    unsigned drop_seed;
} edgedict_stack_layer_t;

#define EDGEDICT_STACK_SERIAL 1     /* run everything on the caller's stream (debug / bit-exact check) */
#define EDGEDICT_STACK_DW_AT_END 2  /* weight gradients after the BPTT instead of under it */
#define EDGEDICT_STACK_ACCUM_GRADS 8 /* dW_ih / dW_hh / db / db_hh are existing gradient buffers: += instead of = */
#define EDGEDICT_STACK_INFERENCE 64 /* no backward pass will follow this forward: the workspace carries no dG images /
   split-K partials (edgedict_stack_workspace_bytes is smaller), edgedict_stack_backward refuses the descriptor */
#define EDGEDICT_STACK_SIDE_STREAM_PER_LAYER 4 /* experiment: chunk GEMMs on one side stream PER LAYER.
   Measured 2x SLOWER end to end: more streams than hardware queues serialises everything. */

typedef struct edgedict_stack_desc {
    int B, H, L;
    int chunk;             /* frames (at the stack's output rate) per input-product chunk */
    int lag;               /* launches between consecutive layers; 0 = chunk*f0 + 5 */
    int split_k;           /* K slices of the (quiet) weight-gradient products: 1, 2, 4 or 8; 0 = 2 */
    int flags;
    float eps;
    const edgedict_stack_layer_t* layers;   /* HOST array [L] */
    const void* x;         /* [B, T0, I0] batch-first encoder input, x_dtype (ED_F32 / ED_BF16) */
    int x_dtype, T0, I0;
    const float* in_gamma; /* f32 [I0] input LayerNorm */
    const float* in_beta;
    float* in_mean;        /* f32 [B*T0] */
    float* in_rstd;
    const float* h0;       /* f32 [L, B, H] nullable (= zeros) */
    const float* c0;
    void* out;             /* bf16 [B, T_out, H] batch-first stack output */
    const void* dout;      /* backward: bf16 [B, T_out, H] */
    float* d_in_gamma;     /* backward: f32 [I0] ACCUMULATED */
    float* d_in_beta;
    void* ws;              /* edgedict_stack_workspace_bytes(desc) bytes, 256-byte aligned */
    size_t ws_bytes;
    /* backward, optional: called on the HOST, synchronously from edgedict_stack_backward, as soon as the
       LAST kernel that accumulates dW_ih / dW_hh / db of `layer` has been enqueued on the auxiliary
       stream (edgedict_aux_stream(2)) - the layers finish top-down, layer L-1 roughly L-1 chunks of
       launches before layer 0.  A data-parallel host orders that layer's gradient all-reduce behind
       the auxiliary stream's position at that moment, so the exchange of layers L-1 .. 1 runs under
       the BPTT of the layers below (reference: DDP's bucketed all-reduce, cli/lightning.py:325-331).
       The callback must not call back into this library. */
    void (*grads_final)(int layer, void* user);
    void* grads_final_user;
} edgedict_stack_desc_t;

/* sizeof(edgedict_stack_layer_t) (which = 0) / sizeof(edgedict_stack_desc_t) (which = 1): lets a
 * foreign-language binding verify its struct mirror */
size_t edgedict_stack_struct_bytes(int which);
/* The library's internal streams of the current device (0 = recurrence, 1 = chunk GEMMs, 2 = the
 * low-priority weight-gradient stream), created on first use.  Host code that wants to run its
 * own off-critical-path work (e.g. the joint network's weight gradients) concurrently should
 * borrow stream 2 instead of creating more streams: beyond the 4 hardware queues HIP multiplexes
 * streams and every cross-stream dependency gets slower. */
void* edgedict_aux_stream(int which);
/* Debug / test aid: which internal streams of the current device still have work enqueued (hipStreamQuery; bit 0
 * recurrence, bit 1 chunk GEMMs, bit 2 auxiliary, bits 3.. lazily created per-layer side streams).  Every entry
 * point joins the streams it used into the caller's stream before it returns, so with the CALLER's stream idle a
 * set bit is work nothing is ordered behind (tests/conftest.py asserts mask == 0 after every GPU test).  Creates
 * nothing. */
int edgedict_streams_busy(unsigned* mask);
size_t edgedict_stack_workspace_bytes(const edgedict_stack_desc_t* desc);
/* fp32 master weights of one layer (nn.LSTM's weight_ih_l0 [4H, I], weight_hh_l0 [4H, H], biases [4H]) -> the bf16
 * images edgedict_stack_layer_t points at.  Every output is optional (null = skip): wih_p (+ bias_p, needs b_ih / b_hh)
 * and whh_f are what the forward pass reads; wih_t, whh_b (and edgedict_stack_pack_sk's image) only the backward pass -
 * a trainer rebuilds the first group in front of the next forward pass and the second behind it. */
int edgedict_stack_pack_weights(const float* w_ih, const float* w_hh, const float* b_ih,
                                const float* b_hh, int H, int I, void* wih_p, void* wih_t,
                                float* bias_p, void* whh_f, void* whh_b, void* stream);
/* W_hh [4H, H] f32 (H % 64 == 0, H <= 1024) -> bf16 split-K image for the weights-stationary BPTT kernel
 * (edgedict_stack_layer_t.whh_s, 4H*H elements): workgroup (unit block of 64, quarter of the 4H interleaved gate
 * columns) keeps its 128 KB in registers for a whole launch */
int edgedict_stack_pack_sk(const float* w_hh, int H, void* whh_s, void* stream);
/* give-up code of the last encoder-stack launch on the current device that ran into one of its bounded
 * in-kernel waits (0 = none since the last call of this function; reading clears it): 5xx / 6xx a chunk flag of
 * the forward / backward pass, 7xx a peer of the launch-persistent forward, 8xx a side stream's counter wait,
 * 9xx a peer of the split-K BPTT (csrc/stack_kernels.hip) */
int edgedict_stack_wsr_error(void);
/* the 3 give-up words behind edgedict_stack_wsr_error of the current device (pinned, mapped host memory):
 * host = 0 -> the DEVICE-visible address (what edgedict_adam_step_guarded takes as skip_words, n_skip = 3),
 * host = 1 -> the host address of the same words (tests poke it).  NULL on failure. */
void* edgedict_stack_error_words(int host);
/* debug: a zeroed device buffer of >= 8 x 8 int64 in which the first workgroup of every layer of the
 * launch-persistent forward / split-K BPTT launches accumulates the 100 MHz ticks it spent per phase
 * (tools/lpw_trace.py, tools/sk_trace.py); NULL switches it off.  Not part of the hot path. */
int edgedict_stack_wsr_set_trace(void* device_buffer);
int edgedict_stack_forward(const edgedict_stack_desc_t* desc, void* stream);
/* measurement aid: HIP-event time (on the recurrence stream) from the first to the last wavefront
 * launch of the most recent forward (backward = 0) or backward (1) call on this device, and the
 * number of launches in it; blocks until that call's launches have executed. */
int edgedict_stack_last_timing(int backward, float* ms, int* launches);
/* which recurrence kernel the most recent forward (backward = 0) / backward (1) call on this device ran:
 * kind 0 = one launch per time step (stack_fwd_kernel / stack_bwd_kernel), 1 = launch-persistent forward
 * (stack_fwd_lpw_kernel), 2 = split-K weights-stationary BPTT (stack_bwd_sk_kernel); steps_per_launch =
 * consecutive time steps of a layer one launch carries (0 for kind 0). */
int edgedict_stack_last_mode(int backward, int* kind, int* steps_per_launch);
/* measurement aid, opt-in: with on != 0 every wavefront launch of the following forward / backward calls on
 * this device stamps its first workgroup's start and its last workgroup's end (constant 100 MHz clock) into
 * an internal device buffer (64 KB per direction, allocated on first use).  edgedict_stack_launch_times then
 * returns the SUM of the launches' own durations of the most recent call and their number: the kernels'
 * average duration as a profiler's begin/end timestamps see it, without the gaps between dependent launches
 * that edgedict_stack_last_timing's span includes.  The two atomics per workgroup cost time: switch it off
 * for timed runs. */
int edgedict_stack_time_launches(int on);
int edgedict_stack_launch_times(int backward, float* sum_ms, int* launches);
/* debug / profiling: the raw stamps behind edgedict_stack_launch_times - out[2 k] = first workgroup's start,
 * out[2 k + 1] = last workgroup's end of wavefront launch k of the last call (100 MHz ticks), in issue order;
 * synchronises the device.  tools/lpw_timeline.py */
int edgedict_stack_launch_stamps(int backward, unsigned long long* out, int max_launches, int* launches);
/* Dry run of the scheduler (no device needed, nothing is launched; buffer pointers in the descriptor
 * only have to be non-NULL): the launch index that carries every layer-step and the number of
 * launches issued when each chunk's side-stream product was enqueued.
 *   step_launch     HOST int32 [sum_l T_l], layer-major, frame t of layer l at offset(l) + t
 *   chunk_enqueued  HOST int32 [sum_l nchunks_l], nchunks_l = ceil(T_l / (chunk * f_l)); -1 = from the start
 *   n_launches, max_slots  HOST int32: launches of the step kernel, most layer-steps in one launch */
int edgedict_stack_schedule(const edgedict_stack_desc_t* d, int backward, int32_t* step_launch,
                            int32_t* chunk_enqueued, int32_t* n_launches, int32_t* max_slots);
int edgedict_stack_backward(const edgedict_stack_desc_t* desc, void* stream);

/* ------------------------------------------------------------------------------------
 * Streaming helpers.
 * cast / transpose: weight copies in the compute dtype (fp32 master -> bf16), dst[c][r]=src[r][c].
 * colsum: out[n] += sum_m x[m*ld+n]  (bias gradients; ATOMIC accumulate into fp32 out).
 * embedding: nn.Embedding(V,E,padding_idx=PAD) lookup with the BOS left-pad of
 *            Decoder.forward (rnnt/models.py:150-153) folded in; backward scatters with atomics
 *            and skips the padding row.
 * joint_hidden: hid[b,t,u,:] = tanh(E1[b,t,:] + D1[b,u,:])  (Joint.forward rnnt/models.py:169-179
 *            with the first Linear split over [enc;dec]); backward returns fp32
 *            dE1[B,T,J] = sum_u dpre, dD1[B,U1,J] = sum_t dpre with dpre = dhid*(1-hid^2).
 */
int edgedict_cast(int src_dtype, const void* src, int dst_dtype, void* dst, long long n,
                  void* stream);
int edgedict_transpose(int src_dtype, const void* src, int dst_dtype, void* dst, int R, int C,
                       void* stream);
int edgedict_colsum(int dtype, const void* x, long long ld, float* out, long long M, int N,
                    void* stream);
int edgedict_embedding_fwd(int out_dtype, int emb_dtype, const int32_t* tokens, int tok_stride,
                           const void* emb, void* out, int B, int Uout, int E, int V,
                           int prepend_bos, int bos, void* stream);
int edgedict_embedding_bwd(int dtype, const int32_t* tokens, int tok_stride, const void* dout,
                           float* demb, int B, int Uout, int E, int V, int prepend_bos, int bos,
                           int pad, void* stream);
/* dropout: y[i] = x[i] * keep(seed, i) / (1-p) with a counter-based mask (nn.Dropout /
 *          nn.LSTM(dropout=p) in training mode, rnnt/models.py:47-53,145-147); the backward pass is
 *          the same call on the incoming gradient with the same seed.  x == y is allowed.
 * spec_mask: zero-fill the SpecAugment time / frequency intervals (TimeMasking / FrequencyMasking,
 *          rnnt/transforms.py:54-146, applied after frame stacking) of a resident feature batch
 *          x fp32 [B,T,F]; t_iv int32 [B][n_t][2], f_iv int32 [B][n_f][2] = half-open [start,end)
 *          intervals over the frame index / the stacked feature index. */
int edgedict_dropout(int dtype, const void* x, void* y, long long n, float p, unsigned seed,
                     void* stream);
int edgedict_spec_mask(float* x, int B, int T, int F, const int32_t* t_iv, int n_t,
                       const int32_t* f_iv, int n_f, void* stream);
int edgedict_joint_hidden_fwd(int dtype, const void* E1, const void* D1, void* hid, int B, int T,
                              int U1, int J, void* stream);
int edgedict_joint_hidden_bwd(int dtype, const void* dhid, const void* hid, float* dE1,
                              float* dD1, int B, int T, int U1, int J, void* stream);
/* packed-lattice forms (see edgedict_rnnt_loss_forward_packed): hid / dhid hold only valid cells;
 * dE1 rows t >= act_lens[b] and dD1 rows u > label_lens[b] come out as zeros. */
int edgedict_joint_hidden_fwd_packed(int dtype, const void* E1, const void* D1, void* hid,
                                     const int32_t* act_lens, const int32_t* label_lens,
                                     const long long* row_offsets, int B, int T, int U1, int J,
                                     void* stream);
int edgedict_joint_hidden_bwd_packed(int dtype, const void* dhid, const void* hid, float* dE1,
                                     float* dD1, const int32_t* act_lens, const int32_t* label_lens,
                                     const long long* row_offsets, int B, int T, int U1, int J,
                                     void* stream);

/* ------------------------------------------------------------------------------------
 * Optimiser step on flat fp32 buffers (torch.optim.Adam semantics, cli/train.py:135-146,268)
 * and the global-norm clip coefficient of clip_grad_norm_ (cli/train.py:262-267).
 *   effective gradient = g * grad_scale_host * (grad_scale ? *grad_scale : 1): the host factor
 *               carries 1/world_size of the data-parallel mean, the nullable device scalar the
 *               clip coefficient (no host sync);  p_bf16: nullable bf16 copy of the new params
 *   grad_clip_coef: coef = min(1, max_norm / (pre_scale*||g||_2 + 1e-6)), norm_out nullable
 */
int edgedict_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr,
                       float beta1, float beta2, float eps, int step, float weight_decay,
                       float grad_scale_host, const float* grad_scale, void* p_bf16, void* stream);
/* The same step, skipped entirely (p, m, v untouched) when any of the n_skip (<= 16) device-visible words
 * at skip_words is non-zero: the trainer passes edgedict_stack_error_words(0), so the gradients of a step
 * whose encoder stack gave up a bounded in-kernel wait are never applied - without a host synchronisation;
 * the host raises at its next edgedict_stack_wsr_error().  The words live in pinned host memory: a one-lane
 * kernel ORs them into guard_scratch[0] (caller-owned DEVICE word) right before the update, which reads
 * only that word (reading the host words from every workgroup cost 0.3-1 ms per step). */
int edgedict_adam_step_guarded(float* p, const float* g, float* m, float* v, long long n, float lr,
                               float beta1, float beta2, float eps, int step, float weight_decay,
                               float grad_scale_host, const float* grad_scale, void* p_bf16,
                               const unsigned* skip_words, int n_skip, unsigned* guard_scratch,
                               void* stream);
int edgedict_grad_clip_coef(const float* g, long long n, float max_norm, float pre_scale,
                            float* sumsq_ws, float* coef, float* norm_out, void* stream);
/* measurement aid, not on the hot path: a stand-in for a collective library's resident kernels - `workgroups`
 * workgroups of 512 threads with >= 128 live registers per lane that read-modify-write buf[0, n_floats) `passes`
 * times (values unchanged).  tools/rccl_footprint.py runs it on the auxiliary stream at every grads_final point of
 * the backward pass (where cli/lightning.py:325-331's bucketed all-reduce would run) to see what the recurrence
 * launches lose to it and that no bounded in-kernel wait gives up. */
int edgedict_debug_footprint(float* buf, long long n_floats, int workgroups, int passes, void* stream);

/* ------------------------------------------------------------------------------------
 * Log-mel filterbank front-end.  Replaces FilterbankFeatures.forward (rnnt/features.py:106-152,
 * twin parts/features.py:298-347) + Downsample.forward (rnnt/transforms.py:38-51) in one kernel;
 * the STFT spectrum never reaches HBM.
 *   wave      fp32 [B, N] (row stride wave_stride); lengths nullable int32 [B] valid samples
 *   window    fp32 [n_fft]  hann(win_length, periodic=False) centred in n_fft, zeros outside
 *             [win_lo, win_hi);  twiddle fp32 [n_fft/2][2] = cos,sin(2*pi*k/n_fft)
 *   fb        fp32 [n_mels, n_fft/2+1] mel weights; fb_range int32 [n_mels][2] non-zero support
 *   out[b*o_b + (f/stack)*o_group + (f%stack)*o_k + m*o_m], f < frames_out (multiple of stack):
 *             frames f >= 1+N_b/hop (no such STFT frame) and f >= ceil(N_b/hop) (the reference's
 *             seq_len mask) are written as zeros.
 * dither: wave[b,n] += amplitude * N(0,1) in place (rnnt/features.py:111-112), counter-based RNG.
 */
int edgedict_dither(float* wave, long long wave_stride, int B, int N, const int32_t* lengths,
                    float amplitude, unsigned seed, void* stream);
int edgedict_fbank_forward(const float* wave, long long wave_stride, int B, int N,
                           const int32_t* lengths, const float* window, const float* twiddle,
                           const float* fb, const int32_t* fb_range, int n_fft, int win_lo,
                           int win_hi, int hop, int n_mels, float preemph, int do_log, void* out,
                           int out_dtype, long long o_b, long long o_group, long long o_k,
                           long long o_m, int stack, int frames_out, void* stream);
/* The Jasper-derived twin, FilterbankFeatures.forward(x, seq_len) of parts/features.py:298-347:
 * the STFT runs over the whole padded row (reflect padding at N, not at the utterance end),
 * seq_len (samples, int32 [B], nullable) only MASKS frames f >= ceil(seq_len/hop) to zero;
 *   out fp32 [b*o_b + (c*n_mels + m)*o_m + f], c < copies ("frame splicing" as the reference writes
 *   it, :111-123, stacks `copies` identical blocks along the feature axis; o_copy = n_mels*o_m),
 *   frames_out <= o_m (a larger o_m leaves the caller's pad_to padding untouched: pre-zeroed);
 *   normalize 0 none / 1 per_feature / 2 all_features = normalize_batch (:80-109): mean and
 *   unbiased std over the first ceil(seq_len/hop) frames, std + 1e-5;
 *   n_signal (0 or N: none): samples [n_signal, N) are zeros appended AFTER the pre-emphasis - the
 *   reference zero-pads the already pre-emphasised short input to win_length (:289-294). */
int edgedict_fbank_forward_masked(const float* wave, long long wave_stride, int B, int N,
                                  const int32_t* seq_len, const float* window,
                                  const float* twiddle, const float* fb, const int32_t* fb_range,
                                  int n_fft, int win_lo, int win_hi, int hop, int n_mels,
                                  float preemph, int do_log, float* out, long long o_b,
                                  long long o_m, int frames_out, int copies, long long o_copy,
                                  int normalize, int n_signal, void* stream);

/* ------------------------------------------------------------------------------------
 * Batched greedy / streaming search: the whole per-frame loop of Transducer.greedy_decode
 * (rnnt/models.py:243-269) or PytorchStreamDecoder.decode (rnnt/stream.py:102-119) in one call.
 *   E1        dtype, frame t of row b at E1[b*e_row_stride + t*e_frame_stride + j]: the encoder
 *             half of the joint's first Linear, enc_out * W1[:, :P_enc]^T, for all T frames
 *   W1d       dtype [J, P2] view (leading dimension ldw1) = W1[:, P_enc:];  b1 fp32 [J]
 *   W2        dtype [V, J]; b2 fp32 [V];  emb [V, E] (emb_dtype);  Wp dtype [P2, H]; bp fp32 [P2]
 *   w_ih/w_hh/b_ih/b_hh  HOST arrays of L device pointers (prediction-network LSTM layers)
 *   h_state, c_state fp32 [L,B,H] and dec_out dtype [B,P2]: prediction-network state, in/out
 *   unk < 0 : greedy mode  (log-softmax max; score[b] += -max log p; nullable score)
 *   unk >= 0: stream mode  (raw-logit arg-max with the reference's <unk> rule)
 *   tokens_out int32, token of row b at frame t stored at tokens_out[b*tok_stride + t] (nullable)
 *   workspace: edgedict_greedy_workspace_bytes(...) bytes, 256-byte aligned
 */
size_t edgedict_greedy_workspace_bytes(int dtype, int B, int J, int V, int E, int L, int H, int P2);
int edgedict_greedy_decode(int dtype, const void* E1, long long e_row_stride,
                           long long e_frame_stride, int B, int T, int J, const void* W1d,
                           long long ldw1, const float* b1, int P2, const void* W2,
                           const float* b2, int V, const void* emb, int emb_dtype, int E, int L,
                           const void* const* w_ih, const void* const* w_hh,
                           const float* const* b_ih, const float* const* b_hh, int H,
                           const void* Wp, const float* bp, float* h_state, float* c_state,
                           void* dec_out, int blank, int unk, int32_t* tokens_out, int tok_stride,
                           float* score, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------
 * GRU time recurrence (the module_type='GRU' encoder variant, ResLayerNormGRU rnnt/models.py:77-116;
 * PyTorch nn.GRU semantics, gate order r,z,n).  Batch-first buffers, one kernel per time step:
 *   G      [B,T,3H] dtype  in : x W_ih^T + b_ih for every t;  out: r, z, n (post-activation)
 *   Hprev  [B,T,H]  dtype  out: h_{t-1} per step (row t=0 <- h0 or zeros)
 *   Y      [B,T,H]  dtype  out: h_t;   HN [B,T,H] dtype out: W_hn h_{t-1} + b_hn (saved for BPTT)
 *   Whh    [3H,H]   dtype; b_hh fp32 [3H]; h0 fp32 [B,H] nullable; hN fp32 [B,H] nullable
 * backward (after forward, same buffers):
 *   G      in: r,z,n;  out: dL/d(input-side pre-activations)    -> dX, dW_ih, db_ih
 *   DH     [B,T,3H] dtype out: dL/d(hidden-side pre-activations) -> dW_hh (with Hprev), db_hh
 *   dY     [B,T,H] dtype nullable;  WhhT [H,3H] dtype;  dh_ws fp32 [B,H] scratch
 * Limits: H % 8 == 0.  The gradient wrt h0 is not produced.
 */
int edgedict_gru_forward(int dtype, void* G, void* Hprev, void* Y, void* HN, const void* Whh,
                         const float* b_hh, const float* h0, float* hN, int B, int T, int H,
                         void* stream);
int edgedict_gru_backward(int dtype, void* G, void* DH, const void* dY, const void* Hprev,
                          const void* HN, const void* WhhT, float* dh_ws, int B, int T, int H,
                          void* stream);

/* ------------------------------------------------------------------------------------
 * Convolutional waveform front-end (FrontEnd, rnnt/models.py:313-365; DilatedConvBlock :323-339;
 * CausalConv1d :314-318), channels-last activations [B, T, C].  A convolution = im2col gather +
 * edgedict_gemm against the weight viewed as [C_out, C_in*k] (column c*k + j); its backward =
 * two GEMMs + the col2im gather.
 *   conv_out_frames(Tin, k, s)   frames after Conv1d(padding k-1, stride s) minus the k-1 dropped ones
 *   conv_im2col   x [B,Tin,C] (in_dtype) -> cols [B,Tout,C*k] (out_dtype); f32->f32, f32->bf16, bf16->bf16
 *   conv_col2im   dcols [B,Tout,C*k] -> dx [B,Tin,C] (same dtype), gather form (no atomics)
 *   gelu_groupnorm_fwd  out = GroupNorm_1group(GELU(y)) with per-channel gamma/beta; y,out [B,T,C];
 *                       mean, rstd fp32 [B] (saved for backward); exact-erf GELU, eps as nn.GroupNorm
 *   gelu_groupnorm_bwd  dy wrt y, dgamma/dbeta fp32 [C] (written, not accumulated); C <= 256;
 *                       workspace of gelu_groupnorm_bwd_workspace_bytes(B,T,C) bytes
 */
int edgedict_conv_out_frames(int Tin, int k, int s);
int edgedict_conv_im2col(int in_dtype, int out_dtype, const void* x, void* cols, int B, int Tin, int C,
                         int k, int s, void* stream);
int edgedict_conv_col2im(int dtype, const void* dcols, void* dx, int B, int Tin, int C, int k, int s,
                         void* stream);
int edgedict_gelu_groupnorm_fwd(int dtype, const void* y, const float* gamma, const float* beta,
                                void* out, float* mean, float* rstd, int B, int T, int C, float eps,
                                void* stream);
size_t edgedict_gelu_groupnorm_bwd_workspace_bytes(int B, int T, int C);
int edgedict_gelu_groupnorm_bwd(int dtype, const void* y, const void* dout, const float* gamma,
                                const float* mean, const float* rstd, void* dy, float* dgamma,
                                float* dbeta, void* workspace, int B, int T, int C, void* stream);

/* Number of products edgedict_gemm has handed to hipBLASLt in this process (only the large,
 * short-K bf16 NT product of the joint's logits qualifies; csrc/blaslt.cpp).  0 when the vendor
 * library is absent or EDGEDICT_BLASLT=0: every product then runs on this library's kernels. */
long long edgedict_blaslt_calls(void);

/* ------------------------------------------------------------------------------------
 * Streaming encoder step (bf16): PytorchStreamDecoder.decode's `self.encoder(xs, (enc_h, enc_c))`
 * (rnnt/stream.py:93-100) on a chunk of a few frames for B concurrent streams - Encoder.forward
 * rnnt/models.py:131-136 without its output projection: input LayerNorm, then per layer T fused LSTM steps
 * (input product + recurrent product + cell in one launch per frame) and the residual / LayerNorm / TimeReduction
 * of ResLayerNormLSTM.forward :55-75.  One native call instead of ~4 launches per layer-frame plus host glue.
 *   xs            [B, T, I0] x_dtype (ED_F32 or ED_BF16), I0 % 8 == 0;  H % 32 == 0
 *   w_ih / w_hh   HOST arrays [L] of bf16 DEVICE matrices [4H, K_l] / [4H, H] (PyTorch gate order i,f,g,o)
 *   b_ih / b_hh / ln_gamma / ln_beta  HOST arrays [L] of fp32 DEVICE vectors;  reduce HOST int [L] in {1, 2}
 *   h_state / c_state  fp32 [L, B, H], updated in place;  out bf16 [B, T_out, H], T_out returned through *T_out
 */
size_t edgedict_stream_encoder_workspace_bytes(int B, int T, int I0, int H, int L);
int edgedict_stream_encoder_step(const void* xs, int x_dtype, int B, int T, int I0, int H, int L,
                                 const float* in_gamma, const float* in_beta, const void* const* w_ih,
                                 const void* const* w_hh, const float* const* b_ih, const float* const* b_hh,
                                 const float* const* ln_gamma, const float* const* ln_beta, const int* reduce,
                                 float* h_state, float* c_state, void* out, int* T_out, void* workspace,
                                 void* stream);

/* ------------------------------------------------------------------------------------
 * Batched beam search: the reference's legacy Transducer.beam_search (models.py:121-202; Sequence
 * models.py:212-224) for B utterances in lockstep, over the maintained model's prediction network and
 * joint.  Operands as edgedict_greedy_decode, plus
 *   lens_host        HOST int32 [B]: encoder frames of each utterance (<= T)
 *   bos              the token the empty hypothesis feeds first (BOS = 2, zero state)
 *   W                beam width;  max_expansions >= W: cap on pops per utterance and frame
 *                    (exceeding it is an error, never a silent truncation)
 *   tokens_host      HOST int32 [B, max_tokens]: tokens of B[0] (no blanks, no BOS)
 *   ntokens_host     HOST int32 [B];  score_host HOST fp64 [B] = -log p of that hypothesis
 *   prefix           0 / 1: the `prefix` argument of the reference method.  1 = models.py:145-161: at the start of every
 *                    frame the probability of reaching a hypothesis A[j] through a later list entry A[i] that is a
 *                    proper prefix of it (the missing tokens emitted on this frame, each from the prediction-network
 *                    output stored when that position was expanded - Sequence.g) is folded into logp(A[j]) with
 *                    log_aplusb, pairs in the reference's order.  The list logic runs on the host (the call already
 *                    synchronises every lockstep iteration), the joint evaluations as one batched device pass per frame
 *   expansions_host  HOST, nullable: total number of prediction-network steps (pops; with prefix = 1 also one per
 *                    merged pair, as the reference spends them)
 * Scores are accumulated in fp64 from fp32 log-softmax values, as the reference's Python floats
 * are.  The call synchronises the stream (the loop's stop test is data dependent).
 */
size_t edgedict_beam_workspace_bytes(int dtype, int B, int T, int J, int V, int E, int L, int H,
                                     int P2, int W, int max_expansions, int prefix);
int edgedict_beam_search(int dtype, const void* E1, long long e_row_stride,
                         long long e_frame_stride, int B, int T, const int32_t* lens_host, int J,
                         const void* W1d, long long ldw1, const float* b1, int P2, const void* W2,
                         const float* b2, int V, const void* emb, int emb_dtype, int E, int L,
                         const void* const* w_ih, const void* const* w_hh,
                         const float* const* b_ih, const float* const* b_hh, int H,
                         const void* Wp, const float* bp, int blank, int bos, int W,
                         int max_expansions, int prefix, int32_t* tokens_host, int max_tokens,
                         int32_t* ntokens_host, double* score_host, long long* expansions_host,
                         void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EDGEDICT_HIP_H */
