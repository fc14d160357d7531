// Batched greedy / streaming RNN-T search loop for gfx950.
//
// Replaces the per-frame Python loops of
//   Transducer.greedy_decode          rnnt/models.py:243-269  (log_softmax -> max, one symbol per
//                                     encoder frame, prediction net advanced for every row and the
//                                     new state kept only where the symbol is not blank)
//   PytorchStreamDecoder.decode       rnnt/stream.py:102-119  (argmax on raw logits, an arg-max
//                                     that is the <unk> id has its logit set to 0 and the arg-max
//                                     is retaken, prediction net advanced only on non-blank)
// The whole time loop runs inside ONE C call: per frame it enqueues the small dense products on
// the MFMA GEMM (gemm.hip), the LSTM cell (lstm.hip) and three tiny kernels from this file
// (broadcast-add+tanh on one frame, row-wise pick, masked state commit).  Nothing returns to the
// host between frames; token ids and scores stay on the device until the caller reads them.
#include "common.hpp"

#include <vector>

// decode_fused.hip: a frame as five fused launches (bf16 and fp32)
bool ed_decode_fused_ok(int dtype, int emb_dtype, int J, int V, int E, int H, int P2);
size_t ed_decode_fused_ws_bytes(int B, int V);
int ed_decode_fused_frame(int dtype, const void* E1t, long long e_row_stride, int B, int J, const void* W1d,
                          long long ldw1, const float* b1, int P2, const void* W2, const float* b2, int V,
                          const void* emb, int emb_dtype, int E, int L, const void* const* w_ih,
                          const void* const* w_hh, const float* const* b_ih, const float* const* b_hh, int H,
                          const void* Wp, const float* bp, float* h_state, float* c_state, void* dec_out, int blank,
                          int unk, int32_t* tokens_out, long long tok_stride, int t, float* score, void* hid, void* parts,
                          int32_t* pred, float* h_new, float* c_new, void* Y0, void* Y1, hipStream_t s);
int ed_decode_fused_beam_step(int dtype, const void* E1t, long long e_row_stride, int B, int J, const void* W1d,
                              long long ldw1, const float* b1, int P2, const void* W2, const float* b2, int V,
                              const void* emb, int emb_dtype, int E, int L, const void* const* w_ih,
                              const void* const* w_hh, const float* const* b_ih, const float* const* b_hh, int H,
                              const void* Wp, const float* bp, const float* h_state, const float* c_state,
                              const int32_t* pred, void* dec_new, void* hid, float* logits, float* h_new, float* c_new,
                              void* Y0, void* Y1, hipStream_t s);

namespace {

// hid[b, :] = tanh(E1[b*e_stride + :] + D1[b, :])
template <typename T>
__global__ void add_tanh_rows(const T* __restrict__ E1, long long e_stride,
                              const T* __restrict__ D1, T* __restrict__ hid, int B, int J) {
    const long long n = (long long)B * J;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / J), j = (int)(i % J);
        ElemIO<T>::store(hid + i, tanhf(ElemIO<T>::load(E1 + (long long)b * e_stride + j) +
                                        ElemIO<T>::load(D1 + i)));
    }
}

// One wave64 per row of fp32 logits [B, V]:
//   greedy mode (unk < 0): pred = argmax (first index on ties, as torch.max), score -= log p(pred)
//   stream mode (unk >= 0): pred = argmax of raw logits; if pred == unk the logit is treated as 0
//                           and the arg-max is retaken (rnnt/stream.py:105-108)
__global__ __launch_bounds__(256) void pick_kernel(const float* __restrict__ logits, int B, int V,
                                                   int unk, int32_t* __restrict__ pred,
                                                   int32_t* __restrict__ tokens, int tok_stride,
                                                   int t, float* __restrict__ score, int blank) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* z = logits + (long long)b * V;
    float best = -INFINITY;
    int arg = 0x7fffffff;
    for (int v = lane; v < V; v += 64) {
        const float x = z[v];
        if (x > best) { best = x; arg = v; }   // strict >: keeps the first index within a lane
    }
    // wave arg-max with lowest-index tie break
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int oa = __shfl_xor(arg, off, 64);
        if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if (unk >= 0 && arg == unk) {
        best = -INFINITY;
        arg = 0x7fffffff;
        for (int v = lane; v < V; v += 64) {
            const float x = (v == unk) ? 0.f : z[v];
            if (x > best) { best = x; arg = v; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ob = __shfl_xor(best, off, 64);
            const int oa = __shfl_xor(arg, off, 64);
            if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
        }
    }
    if (score) {
        float s = 0.f;
        for (int v = lane; v < V; v += 64) s += expf(z[v] - best);
        s = wave_sum(s);
        if (lane == 0) score[b] += logf(s);  // -(max log p) = log sum exp(z - max)
    }
    // a row without one comparable logit (all NaN / all -inf: `x > best` never held) leaves the sentinel: it emits blank,
    // as the fused frame does (decode_fused.hip dec_lstm_step) - the prediction network of that row does not advance and
    // no out-of-range id reaches `pred`, `tokens` or the embedding gather
    if ((unsigned)arg >= (unsigned)V) arg = min(max(blank, 0), V - 1);
    if (lane == 0) {
        pred[b] = arg;
        if (tokens) tokens[(long long)b * tok_stride + t] = arg;
    }
}

// where pred[b] != blank: dec_out[b] <- dec_new[b], h[l,b] <- h_new[l,b], c[l,b] <- c_new[l,b]
template <typename T>
__global__ void commit_kernel(const int32_t* __restrict__ pred, int blank, T* __restrict__ dec_out,
                              const T* __restrict__ dec_new, int P2, float* __restrict__ h,
                              const float* __restrict__ h_new, float* __restrict__ c,
                              const float* __restrict__ c_new, int L, int B, int H) {
    const long long n_dec = (long long)B * P2;
    const long long n_st = (long long)L * B * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_dec + n_st;
         i += (long long)gridDim.x * blockDim.x) {
        if (i < n_dec) {
            const int b = (int)(i / P2);
            if (pred[b] != blank) dec_out[i] = dec_new[i];
        } else {
            const long long k = i - n_dec;
            const int b = (int)((k / H) % B);
            if (pred[b] != blank) {
                h[k] = h_new[k];
                c[k] = c_new[k];
            }
        }
    }
}

inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

struct Ws {
    size_t D1, hid, logits, pred, x, G, Hprev, Y0, Y1, Cst, h_new, c_new, dec_new, total;
};
inline Ws ws_layout(int esz, int B, int J, int V, int E, int L, int H, int P2) {
    Ws w;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += align256(bytes); return r; };
    w.D1 = take((size_t)B * J * esz);
    w.hid = take((size_t)B * J * esz);
    w.logits = take((size_t)B * V * 4);
    w.pred = take((size_t)B * 4);
    w.x = take((size_t)B * E * esz);
    w.G = take((size_t)B * 4 * H * esz);
    w.Hprev = take((size_t)B * H * esz);
    w.Y0 = take((size_t)B * H * esz);
    w.Y1 = take((size_t)B * H * esz);
    w.Cst = take((size_t)B * H * 4);
    w.h_new = take((size_t)L * B * H * 4);
    w.c_new = take((size_t)L * B * H * 4);
    w.dec_new = take((size_t)B * P2 * esz);
    w.total = o;
    return w;
}

}  // namespace

extern "C" size_t edgedict_greedy_workspace_bytes(int dtype, int B, int J, int V, int E, int L,
                                                  int H, int P2) {
    if (B <= 0) return 0;
    return ws_layout(dtype == ED_F32 ? 4 : 2, B, J, V, E, L, H, P2).total;
}

extern "C" int edgedict_greedy_decode(
    int dtype, const void* E1, long long e_row_stride, long long e_frame_stride, int B, int T,
    int J, const void* W1d, long long ldw1, const float* b1, int P2, const void* W2,
    const float* b2, int V, const void* emb, int emb_dtype, int E, int L,
    const void* const* w_ih, const void* const* w_hh, const float* const* b_ih,
    const float* const* b_hh, int H, const void* Wp, const float* bp, float* h_state,
    float* c_state, void* dec_out, int blank, int unk, int32_t* tokens_out, int tok_stride,
    float* score, void* workspace, void* stream_) {
    ED_CHECK_ARG(dtype == ED_F32 || dtype == ED_BF16, "greedy_decode: bad dtype");
    ED_CHECK_ARG(B > 0 && T >= 0 && J > 0 && V > 0 && L > 0 && H > 0 && P2 > 0 && E > 0,
                 "greedy_decode: bad shape");
    ED_CHECK_ARG(E1 && W1d && b1 && W2 && b2 && emb && w_ih && w_hh && b_ih && b_hh && Wp && bp &&
                     h_state && c_state && dec_out && workspace,
                 "greedy_decode: null pointer");
    hipStream_t s = (hipStream_t)stream_;
    const int esz = dtype == ED_F32 ? 4 : 2;
    const Ws w = ws_layout(esz, B, J, V, E, L, H, P2);
    char* p = (char*)workspace;
    void* D1 = p + w.D1;
    void* hid = p + w.hid;
    float* logits = (float*)(p + w.logits);
    int32_t* pred = (int32_t*)(p + w.pred);
    void* x = p + w.x;
    void* G = p + w.G;
    void* Hprev = p + w.Hprev;
    void* Y[2] = {p + w.Y0, p + w.Y1};
    float* Cst = (float*)(p + w.Cst);
    float* h_new = (float*)(p + w.h_new);
    float* c_new = (float*)(p + w.c_new);
    void* dec_new = p + w.dec_new;

    // a frame is five fused launches (decode_fused.hip; the slice partials live in the logits buffer, which the fused
    // frame never fills); odd shapes: composed from the general kernels below
    const bool fused = ed_decode_fused_ok(dtype, emb_dtype, J, V, E, H, P2) &&
                       align256((size_t)B * V * 4) >= ed_decode_fused_ws_bytes(B, V);
    for (int t = 0; t < T; ++t) {
        int rc;
        if (fused) {
            const char* e1f = (const char*)E1 + (size_t)t * e_frame_stride * esz;
            if ((rc = ed_decode_fused_frame(dtype, e1f, e_row_stride, B, J, W1d, ldw1, b1, P2, W2, b2, V, emb, emb_dtype, E, L,
                                            w_ih, w_hh, b_ih, b_hh, H, Wp, bp, h_state, c_state, dec_out, blank, unk,
                                            tokens_out, tok_stride, t, score, hid, logits, pred, h_new, c_new, Y[0],
                                            Y[1], s)))
                return rc;
            continue;
        }
        // joint on frame t:  D1 = dec_out W1d^T + b1;  hid = tanh(E1[:,t] + D1);  logits (fp32)
        if ((rc = edgedict_gemm(dtype, dtype, dec_out, P2, 1, W1d, ldw1, 1, D1, J, B, J, P2, b1,
                                nullptr, 0, 1, s)))
            return rc;
        const char* e1t = (const char*)E1 + (size_t)t * e_frame_stride * esz;
        if (dtype == ED_F32)
            hipLaunchKernelGGL(add_tanh_rows<float>, dim3(ed_grid_for((long long)B * J, 256)),
                               dim3(256), 0, s, (const float*)e1t, e_row_stride, (const float*)D1,
                               (float*)hid, B, J);
        else
            hipLaunchKernelGGL(add_tanh_rows<bf16_t>, dim3(ed_grid_for((long long)B * J, 256)),
                               dim3(256), 0, s, (const bf16_t*)e1t, e_row_stride,
                               (const bf16_t*)D1, (bf16_t*)hid, B, J);
        if ((rc = edgedict_gemm(dtype, ED_F32, hid, J, 1, W2, J, 1, logits, V, B, V, J, b2, nullptr,
                                0, 1, s)))
            return rc;
        hipLaunchKernelGGL(pick_kernel, dim3((B + 3) / 4), dim3(256), 0, s, logits, B, V, unk, pred,
                           tokens_out, tok_stride, t, score, blank);
        // prediction network step on the picked symbols (for every row, committed where != blank)
        if ((rc = edgedict_embedding_fwd(dtype, emb_dtype, pred, 1, emb, x, B, 1, E, V, 0, 0, s)))
            return rc;
        const void* xin = x;
        int xin_dim = E;
        for (int k = 0; k < L; ++k) {
            if ((rc = edgedict_gemm(dtype, dtype, xin, xin_dim, 1, w_ih[k], xin_dim, 1, G, 4 * H, B,
                                    4 * H, xin_dim, b_ih[k], b_hh[k], 0, 1, s)))
                return rc;
            if ((rc = edgedict_lstm_forward(dtype, G, Hprev, Y[k & 1], Cst, w_hh[k], nullptr,
                                            h_state + (size_t)k * B * H, c_state + (size_t)k * B * H,
                                            h_new + (size_t)k * B * H, c_new + (size_t)k * B * H, B,
                                            1, H, nullptr, s)))
                return rc;
            xin = Y[k & 1];
            xin_dim = H;
        }
        if ((rc = edgedict_gemm(dtype, dtype, xin, H, 1, Wp, H, 1, dec_new, P2, B, P2, H, bp,
                                nullptr, 0, 1, s)))
            return rc;
        const long long n = (long long)B * P2 + (long long)L * B * H;
        if (dtype == ED_F32)
            hipLaunchKernelGGL(commit_kernel<float>, dim3(ed_grid_for(n, 256)), dim3(256), 0, s,
                               pred, blank, (float*)dec_out, (const float*)dec_new, P2, h_state,
                               h_new, c_state, c_new, L, B, H);
        else
            hipLaunchKernelGGL(commit_kernel<bf16_t>, dim3(ed_grid_for(n, 256)), dim3(256), 0, s,
                               pred, blank, (bf16_t*)dec_out, (const bf16_t*)dec_new, P2, h_state,
                               h_new, c_state, c_new, L, B, H);
    }
    ED_CHECK_LAUNCH("greedy_decode");
    return ED_OK;
}


// =====================================================================================
// Beam search (Graves 2012) as the reference's legacy Transducer.beam_search runs it
// (/root/reference/models.py:121-202, prefix=False), for a batch of utterances in lockstep.
//
// Per utterance and frame the reference keeps two Python lists: A (hypotheses still to expand) and
// B (hypotheses that emitted blank in this frame).  Here A is a score POOL in list order:
//   pool[0 .. W)                 the (at most W) survivors of the previous frame
//   pool[W + e*V + k]            child k of the e-th expansion of this frame (-inf for k == blank
//                                and for popped entries)
// so "max(A), first one on ties" is an arg-max with lowest-index tie break over the pool.  A child
// is materialised (token-tree node, state) only when it is popped.  Hypothesis = (node of the token
// tree, prediction-network state reference, last token, fp64 log-probability); the state a
// hypothesis carries is the one BEFORE its last token was consumed (models.py:170-171,185), the
// expansion recomputes the step on that last token.  B keeps its first W entries in insertion order
// (the reference's `sorted(...)` calls discard their result, models.py:141,195-196).
// One lockstep iteration = pop -> prediction-network step -> joint -> expand for every utterance
// whose loop is still open; the host reads the open flags once per iteration after the first W
// (an utterance needs at least W expansions per frame to fill B).
// =====================================================================================
namespace {

struct BeamPtrs {
    double* pool;        // [B][W + EM*V]
    double* bp_logp;     // [B][W]   survivors of the previous frame
    int32_t* bp_node;    // [B][W]
    int32_t* n_bp;       // [B]
    float* bp_h[2];      // [B][W][L][H] double-buffered per frame
    float* bp_c[2];
    double* bn_logp;     // [B][W]   B of the current frame (first W entries)
    int32_t* bn_node;
    int32_t* bn_ref;
    int32_t* n_b;        // [B]      len(B)
    double* max_b;       // [B]
    double* blk_max;     // [B][1 + EM]  max of pool segment 0 (the survivors) / 1 + e (children of expansion e) ...
    int32_t* blk_arg;    // [B][1 + EM]  ... and its first position in the pool (INT_MAX: nothing left)
    int32_t* exp_node;   // [B][EM]  node / state reference / score of the e-th popped hypothesis
    int32_t* exp_ref;
    double* exp_logp;
    int32_t* e_count;    // [B]
    float* f_h;          // [B][EM][L][H] states created in this frame
    float* f_c;
    int2* nodes;         // [B][NODES] (parent, token)
    int32_t* n_nodes;    // [B]
    int32_t* open;       // [B] the while-loop of this utterance is still running
    int32_t* lens;       // [B]
    int32_t* flags;      // [0] expansion cap hit, [1] node cap hit
    long long* total_exp;// [1]
};

// (max, first position) of a workgroup's per-thread candidates: butterfly inside a wave, the four waves through LDS.
// Ties go to the lower position, -inf candidates carry INT_MAX.  Every thread gets the result.
__device__ __forceinline__ void block_argmax(double& v, int& a, double* s_v, int* s_a) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double ov = __shfl_xor(v, off, 64);
        const int oa = __shfl_xor(a, off, 64);
        if (ov > v || (ov == v && oa < a)) { v = ov; a = oa; }
    }
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();                       // (s_v / s_a may still be read from a previous use)
    if ((threadIdx.x & 63) == 0) { s_v[wave] = v; s_a[wave] = a; }
    __syncthreads();
    v = s_v[0]; a = s_a[0];
    for (int w = 1; w < nw; ++w)
        if (s_v[w] > v || (s_v[w] == v && s_a[w] < a)) { v = s_v[w]; a = s_a[w]; }
}
__device__ __forceinline__ float block_sum_max(float x, bool is_max, float* s_f) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float o = __shfl_xor(x, off, 64);
        x = is_max ? fmaxf(x, o) : x + o;
    }
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_f[wave] = x;
    __syncthreads();
    x = s_f[0];
    for (int w = 1; w < nw; ++w) x = is_max ? fmaxf(x, s_f[w]) : x + s_f[w];
    return x;
}

// A pool of W + expansions x V fp64 scores per utterance is searched twice per iteration (arg-max for the pop, max for
// the stop test): 20 k scores after ten expansions.  Each segment - the survivors, the V children of one expansion -
// therefore keeps its (max, first position); a pop or an expansion rescans ONE segment and then only the segment maxima.
__global__ __launch_bounds__(64) void beam_frame_begin(BeamPtrs p, int t, int W, int EM, int V) {
    const int b = blockIdx.x;
    const bool act = t < p.lens[b];
    if (threadIdx.x == 0) {
        p.open[b] = act ? 1 : 0;
        if (act) {
            p.e_count[b] = 0;
            p.n_b[b] = 0;
            p.max_b[b] = -INFINITY;
        }
    }
    if (!act) return;
    double best = -INFINITY;
    int arg = 0x7fffffff;
    const int nbp = p.n_bp[b];
    for (int i = threadIdx.x; i < W; i += 64) {
        const double v = i < nbp ? p.bp_logp[b * W + i] : -INFINITY;
        p.pool[(size_t)b * (W + (size_t)EM * V) + i] = v;
        if (v > best) { best = v; arg = i; }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double ov = __shfl_xor(best, off, 64);
        const int oa = __shfl_xor(arg, off, 64);
        if (ov > best || (ov == best && oa < arg)) { best = ov; arg = oa; }
    }
    if (threadIdx.x == 0) {
        p.blk_max[(size_t)b * (EM + 1)] = best;
        p.blk_arg[(size_t)b * (EM + 1)] = arg;
    }
}

// y* = max(A) (first on ties), removed from A; its last token -> pred, its state -> h/c_state
__global__ __launch_bounds__(256) void beam_pop(BeamPtrs p, int cur, int W, int V, int EM, int L,
                                                int H, int B, int NODES, int bos,
                                                int32_t* __restrict__ pred, float* __restrict__ h_state,
                                                float* __restrict__ c_state) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (!p.open[b]) return;
    double* pool = p.pool + (size_t)b * (W + (size_t)EM * V);
    const int e_n = p.e_count[b];
    double* bmax = p.blk_max + (size_t)b * (EM + 1);
    int32_t* barg = p.blk_arg + (size_t)b * (EM + 1);
    __shared__ double sb[4];
    __shared__ int sa[4];
    __shared__ int s_ref;
    // arg-max over the segment maxima (segments are in pool order, so the first position of the lowest segment wins ties)
    double best = -INFINITY;
    int arg = 0x7fffffff;
    for (int i = tid; i <= e_n; i += 256) {
        const double v = bmax[i];
        const int a = barg[i];
        if (v > best || (v == best && a < arg)) { best = v; arg = a; }
    }
    block_argmax(best, arg, sb, sa);
    const int idx = arg;
    const double popped = best;
    if (idx == 0x7fffffff) {                 // nothing left in A (cannot happen: a frame starts with >= 1 survivor)
        if (tid == 0) { p.flags[0] = 1; p.open[b] = 0; }
        return;
    }
    // the popped entry leaves A: rescan its segment
    {
        const int seg = idx < W ? 0 : 1 + (idx - W) / V;
        const int lo = seg == 0 ? 0 : W + (seg - 1) * V, n = seg == 0 ? W : V;
        double v2 = -INFINITY;
        int a2 = 0x7fffffff;
        for (int i = tid; i < n; i += 256) {
            const double v = (lo + i == idx) ? -INFINITY : pool[lo + i];
            if (v > v2) { v2 = v; a2 = lo + i; }
        }
        block_argmax(v2, a2, sb, sa);
        if (tid == 0) { bmax[seg] = v2; barg[seg] = v2 == -INFINITY ? 0x7fffffff : a2; }
    }
    if (tid == 0) {
        const int e = e_n;
        int node, ref, tok;
        if (idx < W) {
            node = p.bp_node[b * W + idx];
            ref = idx;
        } else {
            const int ep = (idx - W) / V, k = (idx - W) % V;
            int nn = p.n_nodes[b];
            if (nn >= NODES) { p.flags[1] = 1; nn = NODES - 1; }
            p.nodes[(size_t)b * NODES + nn] = make_int2(p.exp_node[b * EM + ep], k);
            p.n_nodes[b] = nn + 1;
            node = nn;
            ref = W + ep;
        }
        tok = node < 0 ? bos : p.nodes[(size_t)b * NODES + node].y;
        pool[idx] = -INFINITY;
        p.exp_node[b * EM + e] = node;
        p.exp_ref[b * EM + e] = ref;
        p.exp_logp[b * EM + e] = popped;
        pred[b] = tok;
        s_ref = ref;
    }
    __syncthreads();
    const int ref = s_ref;
    const size_t LH = (size_t)L * H;
    const float* sh = ref < W ? p.bp_h[cur] + ((size_t)b * W + ref) * LH : p.f_h + ((size_t)b * EM + (ref - W)) * LH;
    const float* sc = ref < W ? p.bp_c[cur] + ((size_t)b * W + ref) * LH : p.f_c + ((size_t)b * EM + (ref - W)) * LH;
    for (int i = tid; i < (int)LH; i += 256) {
        const int l = i / H, j = i % H;
        h_state[((size_t)l * B + b) * H + j] = sh[i];
        c_state[((size_t)l * B + b) * H + j] = sc[i];
    }
}

// log-softmax of the joint's logits; children into the pool, the blank child into B, the new
// prediction-network state into the frame's state pool; then the reference's stop test
__global__ __launch_bounds__(256) void beam_expand(BeamPtrs p, const float* __restrict__ logits,
                                                   const float* __restrict__ h_new,
                                                   const float* __restrict__ c_new, int W, int V,
                                                   int EM, int L, int H, int B, int blank) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (!p.open[b]) return;
    const float* z = logits + (size_t)b * V;
    __shared__ float sf[4];
    __shared__ double sd[4];
    __shared__ int si[4];
    float m = -INFINITY;
    for (int v = tid; v < V; v += 256) m = fmaxf(m, z[v]);
    m = block_sum_max(m, true, sf);
    float s = 0.f;
    for (int v = tid; v < V; v += 256) s += expf(z[v] - m);
    s = block_sum_max(s, false, sf);
    const float logs = logf(s);
    const int e = p.e_count[b];
    const double base = p.exp_logp[b * EM + e];
    double* pool = p.pool + (size_t)b * (W + (size_t)EM * V);
    const int lo = W + e * V;
    double* seg = pool + lo;
    double best = -INFINITY;
    int arg = 0x7fffffff;
    for (int v = tid; v < V; v += 256) {
        const double c = v == blank ? -INFINITY : base + (double)((z[v] - m) - logs);
        seg[v] = c;
        if (c > best) { best = c; arg = lo + v; }
    }
    const size_t LH = (size_t)L * H;
    float* dh = p.f_h + ((size_t)b * EM + e) * LH;
    float* dc = p.f_c + ((size_t)b * EM + e) * LH;
    for (int i = tid; i < (int)LH; i += 256) {
        const int l = i / H, j = i % H;
        dh[i] = h_new[((size_t)l * B + b) * H + j];
        dc[i] = c_new[((size_t)l * B + b) * H + j];
    }
    // the new segment's (max, first position), then max(A) = max over the segment maxima
    block_argmax(best, arg, sd, si);
    double* bmax = p.blk_max + (size_t)b * (EM + 1);
    if (tid == 0) {
        bmax[e + 1] = best;
        p.blk_arg[(size_t)b * (EM + 1) + e + 1] = arg;
    }
    double amax = best;
    int dummy = 0;
    for (int i = tid; i <= e; i += 256) amax = fmax(amax, bmax[i]);       // (segments 0 .. e: written by earlier kernels)
    block_argmax(amax, dummy, sd, si);
    if (tid == 0) {
        const double lpb = base + (double)((z[blank] - m) - logs);
        const int j = p.n_b[b];
        if (j < W) {
            p.bn_logp[b * W + j] = lpb;
            p.bn_node[b * W + j] = p.exp_node[b * EM + e];
            p.bn_ref[b * W + j] = p.exp_ref[b * EM + e];
        }
        p.n_b[b] = j + 1;
        const double mb = fmax(p.max_b[b], lpb);
        p.max_b[b] = mb;
        p.e_count[b] = e + 1;
        atomicAdd((unsigned long long*)p.total_exp, 1ull);
        if (j + 1 >= W && mb >= amax) {
            p.open[b] = 0;
        } else if (e + 1 >= EM) {
            p.flags[0] = 1;
            p.open[b] = 0;
        }
    }
}

// B = B[:W] becomes the next frame's survivors (states gathered into the other buffer)
__global__ __launch_bounds__(256) void beam_frame_end(BeamPtrs p, int t, int cur, int W, int EM,
                                                      int L, int H) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (t >= p.lens[b]) return;
    const int n = min(p.n_b[b], W);
    const size_t LH = (size_t)L * H;
    for (int j = 0; j < n; ++j) {
        const int ref = p.bn_ref[b * W + j];
        const float* sh = ref < W ? p.bp_h[cur] + ((size_t)b * W + ref) * LH : p.f_h + ((size_t)b * EM + (ref - W)) * LH;
        const float* sc = ref < W ? p.bp_c[cur] + ((size_t)b * W + ref) * LH : p.f_c + ((size_t)b * EM + (ref - W)) * LH;
        float* dh = p.bp_h[cur ^ 1] + ((size_t)b * W + j) * LH;
        float* dc = p.bp_c[cur ^ 1] + ((size_t)b * W + j) * LH;
        for (int i = tid; i < (int)LH; i += 256) { dh[i] = sh[i]; dc[i] = sc[i]; }
    }
    if (tid < n) {
        p.bp_logp[b * W + tid] = p.bn_logp[b * W + tid];
        p.bp_node[b * W + tid] = p.bn_node[b * W + tid];
    }
    if (tid == 0) p.n_bp[b] = n;
}

__global__ void beam_init(BeamPtrs p, int B, int W) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    p.n_bp[b] = 1;               // B = [empty hypothesis], models.py:148
    p.bp_logp[b * W] = 0.0;
    p.bp_node[b * W] = -1;
    p.n_nodes[b] = 0;
    p.open[b] = 0;
    if (b == 0) { p.flags[0] = p.flags[1] = 0; *p.total_exp = 0; }
}

// ---- prefix=True (models.py:145-161): the prediction-network output of every expansion is kept per token-tree node
// (the reference's Sequence.g), and at the start of a frame the host - which owns the list logic of that branch -
// asks for log p(token | frame t, stored prediction) of a batch of (utterance, node, token) queries
template <typename T>
__global__ void beam_store_pred(BeamPtrs p, const T* __restrict__ dec_new, T* __restrict__ node_pred, int EM,
                                int NODES, int P2) {
    const int b = blockIdx.x;
    if (!p.open[b]) return;
    const int node = p.exp_node[b * EM + p.e_count[b]];        // -1 = the empty hypothesis: slot 0
    T* dst = node_pred + ((size_t)b * (NODES + 1) + (node + 1)) * P2;
    for (int i = threadIdx.x; i < P2; i += blockDim.x) dst[i] = dec_new[(size_t)b * P2 + i];
}
template <typename T>
__global__ void beam_query_gather(const T* __restrict__ node_pred, const int32_t* __restrict__ q_row,
                                  const int32_t* __restrict__ q_node, T* __restrict__ q_pred, int NODES, int P2) {
    const int q = blockIdx.x;
    const T* src = node_pred + ((size_t)q_row[q] * (NODES + 1) + (q_node[q] + 1)) * P2;
    for (int i = threadIdx.x; i < P2; i += blockDim.x) q_pred[(size_t)q * P2 + i] = src[i];
}
// hid[q, :] = tanh(E1[row(q), frame t, :] + D1[q, :])
template <typename T>
__global__ void beam_query_hidden(const T* __restrict__ E1t, long long e_stride, const int32_t* __restrict__ q_row,
                                  const T* __restrict__ D1, T* __restrict__ hid, int J) {
    const int q = blockIdx.x;
    const T* e = E1t + (long long)q_row[q] * e_stride;
    for (int j = threadIdx.x; j < J; j += blockDim.x)
        ElemIO<T>::store(hid + (size_t)q * J + j, tanhf(ElemIO<T>::load(e + j) + ElemIO<T>::load(D1 + (size_t)q * J + j)));
}
// out[q] = log_softmax(logits[q, :])[tok(q)], the arithmetic of beam_expand
__global__ __launch_bounds__(256) void beam_query_logp(const float* __restrict__ logits, const int32_t* __restrict__ q_tok,
                                                       float* __restrict__ out, int V) {
    const int q = blockIdx.x, tid = threadIdx.x;
    const float* z = logits + (size_t)q * V;
    __shared__ float sf[256];
    float m = -INFINITY;
    for (int v = tid; v < V; v += 256) m = fmaxf(m, z[v]);
    sf[tid] = m;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) sf[tid] = fmaxf(sf[tid], sf[tid + off]);
        __syncthreads();
    }
    m = sf[0];
    __syncthreads();
    float sm = 0.f;
    for (int v = tid; v < V; v += 256) sm += expf(z[v] - m);
    sf[tid] = sm;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) sf[tid] += sf[tid + off];
        __syncthreads();
    }
    if (tid == 0) out[q] = (z[q_tok[q]] - m) - logf(sf[0]);
}
constexpr int BEAM_QMAX = 1024;      // queries per batch of the prefix branch

struct BeamWs {
    Ws step;   // the greedy loop's per-iteration buffers (prediction-network step + joint)
    size_t h_state, c_state;
    size_t pool, bp_logp, bp_node, n_bp, bp_h0, bp_h1, bp_c0, bp_c1, bn_logp, bn_node, bn_ref, n_b, max_b, blk_max, blk_arg,
        exp_node, exp_ref, exp_logp, e_count, f_h, f_c, nodes, n_nodes, open, lens, flags, total_exp, total;
    size_t node_pred, q_row, q_node, q_tok, q_pred, q_D1, q_hid, q_logits, q_out;      // prefix = 1 only
};
inline BeamWs beam_layout(int esz, int B, int T, int J, int V, int E, int L, int H, int P2, int W, int EM, int prefix) {
    BeamWs w;
    w.step = ws_layout(esz, B, J, V, E, L, H, P2);
    size_t o = w.step.total;
    auto take = [&](size_t bytes) { size_t r = o; o += align256(bytes); return r; };
    const size_t LH = (size_t)L * H, NODES = (size_t)T * EM + 1;
    w.h_state = take((size_t)B * LH * 4);
    w.c_state = take((size_t)B * LH * 4);
    w.pool = take((size_t)B * (W + (size_t)EM * V) * 8);
    w.bp_logp = take((size_t)B * W * 8);
    w.bp_node = take((size_t)B * W * 4);
    w.n_bp = take((size_t)B * 4);
    w.bp_h0 = take((size_t)B * W * LH * 4);
    w.bp_h1 = take((size_t)B * W * LH * 4);
    w.bp_c0 = take((size_t)B * W * LH * 4);
    w.bp_c1 = take((size_t)B * W * LH * 4);
    w.bn_logp = take((size_t)B * W * 8);
    w.bn_node = take((size_t)B * W * 4);
    w.bn_ref = take((size_t)B * W * 4);
    w.n_b = take((size_t)B * 4);
    w.max_b = take((size_t)B * 8);
    w.blk_max = take((size_t)B * (EM + 1) * 8);
    w.blk_arg = take((size_t)B * (EM + 1) * 4);
    w.exp_node = take((size_t)B * EM * 4);
    w.exp_ref = take((size_t)B * EM * 4);
    w.exp_logp = take((size_t)B * EM * 8);
    w.e_count = take((size_t)B * 4);
    w.f_h = take((size_t)B * EM * LH * 4);
    w.f_c = take((size_t)B * EM * LH * 4);
    w.nodes = take((size_t)B * NODES * 8);
    w.n_nodes = take((size_t)B * 4);
    w.open = take((size_t)B * 4);
    w.lens = take((size_t)B * 4);
    w.flags = take(8);
    w.total_exp = take(8);
    w.node_pred = w.q_row = w.q_node = w.q_tok = w.q_pred = w.q_D1 = w.q_hid = w.q_logits = w.q_out = 0;
    if (prefix) {
        w.node_pred = take((size_t)B * (NODES + 1) * P2 * esz);
        w.q_row = take((size_t)BEAM_QMAX * 4);
        w.q_node = take((size_t)BEAM_QMAX * 4);
        w.q_tok = take((size_t)BEAM_QMAX * 4);
        w.q_pred = take((size_t)BEAM_QMAX * P2 * esz);
        w.q_D1 = take((size_t)BEAM_QMAX * J * esz);
        w.q_hid = take((size_t)BEAM_QMAX * J * esz);
        w.q_logits = take((size_t)BEAM_QMAX * V * 4);
        w.q_out = take((size_t)BEAM_QMAX * 4);
    }
    w.total = o;
    return w;
}

}  // namespace

extern "C" size_t edgedict_beam_workspace_bytes(int dtype, int B, int T, int J, int V, int E, int L,
                                                int H, int P2, int W, int max_expansions, int prefix) {
    if (B <= 0 || W <= 0 || max_expansions <= 0) return 0;
    return beam_layout(dtype == ED_F32 ? 4 : 2, B, T < 0 ? 0 : T, J, V, E, L, H, P2, W, max_expansions, prefix).total;
}

extern "C" int edgedict_beam_search(
    int dtype, const void* E1, long long e_row_stride, long long e_frame_stride, int B, int T,
    const int32_t* lens_host, int J, const void* W1d, long long ldw1, const float* b1, int P2,
    const void* W2, const float* b2, int V, const void* emb, int emb_dtype, int E, int L,
    const void* const* w_ih, const void* const* w_hh, const float* const* b_ih,
    const float* const* b_hh, int H, const void* Wp, const float* bp, int blank, int bos, int W,
    int max_expansions, int prefix, int32_t* tokens_host, int max_tokens, int32_t* ntokens_host,
    double* score_host, long long* expansions_host, void* workspace, void* stream_) {
    ED_CHECK_ARG(dtype == ED_F32 || dtype == ED_BF16, "beam_search: bad dtype");
    ED_CHECK_ARG(B > 0 && T >= 0 && J > 0 && V > 0 && L > 0 && H > 0 && P2 > 0 && E > 0 && W > 0 &&
                     max_expansions >= W && max_tokens >= 0,
                 "beam_search: bad shape (max_expansions must be >= W)");
    ED_CHECK_ARG((T == 0 || E1) && W1d && b1 && W2 && b2 && emb && w_ih && w_hh && b_ih && b_hh &&
                     Wp && bp && lens_host && tokens_host && ntokens_host && score_host && workspace,
                 "beam_search: null pointer");
    ED_CHECK_ARG(blank >= 0 && blank < V && bos >= 0 && bos < V, "beam_search: blank/bos outside the vocabulary");
    for (int b = 0; b < B; ++b)
        ED_CHECK_ARG(lens_host[b] >= 0 && lens_host[b] <= T, "beam_search: lens[%d] = %d outside [0, %d]", b, lens_host[b], T);
    hipStream_t s = (hipStream_t)stream_;
    const int esz = dtype == ED_F32 ? 4 : 2, EM = max_expansions;
    const int NODES = T * EM + 1;
    const BeamWs w = beam_layout(esz, B, T, J, V, E, L, H, P2, W, EM, prefix);
    char* p = (char*)workspace;
    void* D1 = p + w.step.D1;
    void* hid = p + w.step.hid;
    float* logits = (float*)(p + w.step.logits);
    int32_t* pred = (int32_t*)(p + w.step.pred);
    void* x = p + w.step.x;
    void* G = p + w.step.G;
    void* Hprev = p + w.step.Hprev;
    void* Y[2] = {p + w.step.Y0, p + w.step.Y1};
    float* Cst = (float*)(p + w.step.Cst);
    float* h_new = (float*)(p + w.step.h_new);
    float* c_new = (float*)(p + w.step.c_new);
    void* dec_new = p + w.step.dec_new;
    float* h_state = (float*)(p + w.h_state);
    float* c_state = (float*)(p + w.c_state);
    BeamPtrs q;
    q.pool = (double*)(p + w.pool);
    q.bp_logp = (double*)(p + w.bp_logp);
    q.bp_node = (int32_t*)(p + w.bp_node);
    q.n_bp = (int32_t*)(p + w.n_bp);
    q.bp_h[0] = (float*)(p + w.bp_h0); q.bp_h[1] = (float*)(p + w.bp_h1);
    q.bp_c[0] = (float*)(p + w.bp_c0); q.bp_c[1] = (float*)(p + w.bp_c1);
    q.bn_logp = (double*)(p + w.bn_logp);
    q.bn_node = (int32_t*)(p + w.bn_node);
    q.bn_ref = (int32_t*)(p + w.bn_ref);
    q.n_b = (int32_t*)(p + w.n_b);
    q.max_b = (double*)(p + w.max_b);
    q.blk_max = (double*)(p + w.blk_max);
    q.blk_arg = (int32_t*)(p + w.blk_arg);
    q.exp_node = (int32_t*)(p + w.exp_node);
    q.exp_ref = (int32_t*)(p + w.exp_ref);
    q.exp_logp = (double*)(p + w.exp_logp);
    q.e_count = (int32_t*)(p + w.e_count);
    q.f_h = (float*)(p + w.f_h);
    q.f_c = (float*)(p + w.f_c);
    q.nodes = (int2*)(p + w.nodes);
    q.n_nodes = (int32_t*)(p + w.n_nodes);
    q.open = (int32_t*)(p + w.open);
    q.lens = (int32_t*)(p + w.lens);
    q.flags = (int32_t*)(p + w.flags);
    q.total_exp = (long long*)(p + w.total_exp);

    const size_t LH = (size_t)L * H;
    ED_CHECK_HIP(hipMemcpyAsync(q.lens, lens_host, (size_t)B * 4, hipMemcpyHostToDevice, s));
    // the empty hypothesis' state is zero (Decoder.forward(empty, None), rnnt/models.py:150-153)
    ED_CHECK_HIP(hipMemsetAsync(q.bp_h[0], 0, (size_t)B * W * LH * 4, s));
    ED_CHECK_HIP(hipMemsetAsync(q.bp_c[0], 0, (size_t)B * W * LH * 4, s));
    // rows of finished / not yet popped utterances flow through the step kernels: keep them finite
    ED_CHECK_HIP(hipMemsetAsync(h_state, 0, (size_t)B * LH * 4, s));
    ED_CHECK_HIP(hipMemsetAsync(c_state, 0, (size_t)B * LH * 4, s));
    ED_CHECK_HIP(hipMemsetAsync(pred, 0, (size_t)B * 4, s));
    hipLaunchKernelGGL(beam_init, dim3((B + 63) / 64), dim3(64), 0, s, q, B, W);
    std::vector<int32_t> open_h(B);
    int maxlen = 0;
    for (int b = 0; b < B; ++b) maxlen = lens_host[b] > maxlen ? lens_host[b] : maxlen;

    // ---- prefix = 1: host mirrors for the list logic of models.py:145-161
    long long prefix_steps = 0;                 // prediction-network steps the reference spends in that branch
    std::vector<int32_t> nbp_h, node_h, nn_h, qrow, qnode, qtok;
    std::vector<double> logp_h;
    std::vector<int2> nodes_h;
    std::vector<float> qout;
    // the token tree is APPEND-ONLY (beam_expand writes node n_nodes[b]++ and nothing else): the host keeps a mirror and
    // fetches, per frame, only the node columns that are new since its last fetch - one strided copy of
    // [B] x [min_b mirrored[b], max_b n_nodes[b]) instead of the whole B x NODES tree (O(T^2 EM B) bytes per utterance
    // batch)
    std::vector<int32_t> mirrored;
    struct Pair { int b, j, i, q0, nq; };
    std::vector<Pair> pairs;
    std::vector<std::vector<int>> paths;                          // per list entry: its nodes, last token first
    if (prefix) { nodes_h.resize((size_t)B * NODES); mirrored.assign(B, 0); paths.resize(W); }
    void* node_pred = prefix ? (void*)(p + w.node_pred) : nullptr;
    auto prefix_merge = [&](int t, const char* e1t) -> int {
        nbp_h.resize(B); node_h.resize((size_t)B * W); logp_h.resize((size_t)B * W); nn_h.resize(B);
        ED_CHECK_HIP(hipMemcpyAsync(nbp_h.data(), q.n_bp, (size_t)B * 4, hipMemcpyDeviceToHost, s));
        ED_CHECK_HIP(hipMemcpyAsync(nn_h.data(), q.n_nodes, (size_t)B * 4, hipMemcpyDeviceToHost, s));
        ED_CHECK_HIP(hipStreamSynchronize(s));
        bool any = false;
        for (int b = 0; b < B; ++b) any = any || (t < lens_host[b] && nbp_h[b] > 1);
        if (!any) return ED_OK;
        int lo = NODES, hi = 0;
        for (int b = 0; b < B; ++b) {
            if (nn_h[b] > mirrored[b]) { lo = std::min(lo, mirrored[b]); hi = std::max(hi, nn_h[b]); }
        }
        ED_CHECK_HIP(hipMemcpyAsync(node_h.data(), q.bp_node, node_h.size() * 4, hipMemcpyDeviceToHost, s));
        ED_CHECK_HIP(hipMemcpyAsync(logp_h.data(), q.bp_logp, logp_h.size() * 8, hipMemcpyDeviceToHost, s));
        if (hi > lo)
            ED_CHECK_HIP(hipMemcpy2DAsync(nodes_h.data() + lo, (size_t)NODES * 8, q.nodes + lo, (size_t)NODES * 8,
                                          (size_t)(hi - lo) * 8, (size_t)B, hipMemcpyDeviceToHost, s));
        ED_CHECK_HIP(hipStreamSynchronize(s));
        for (int b = 0; b < B; ++b) mirrored[b] = std::max(mirrored[b], nn_h[b]);
        // every (j, i > j) with A[i] a proper prefix of A[j]: the tokens of A[j] behind A[i], each with the prediction
        // stored at the node before it on A[j]'s path (g; the first one is the prediction of A[i]'s sequence)
        pairs.clear();
        qrow.clear(); qnode.clear(); qtok.clear();
        for (int b = 0; b < B; ++b) {
            if (t >= lens_host[b]) continue;
            const int n = nbp_h[b];
            const int2* nd = nodes_h.data() + (size_t)b * NODES;
            for (int j = 0; j < n; ++j) {
                paths[j].clear();
                for (int x = node_h[(size_t)b * W + j]; x >= 0; x = nd[x].x) paths[j].push_back(x);
            }
            for (int j = 0; j + 1 < n; ++j) {
                const std::vector<int>& pj = paths[j];
                const int lj = (int)pj.size();
                for (int i = j + 1; i < n; ++i) {
                    // isprefix(A[i].k, A[j].k), by VALUE: the same token sequence can sit in the list twice, as two
                    // nodes (a hypothesis that survived as a blank child AND was re-created from its parent)
                    const std::vector<int>& pi = paths[i];
                    const int li = (int)pi.size();
                    if (li >= lj) continue;
                    bool pre = true;
                    for (int d = 0; d < li && pre; ++d) pre = nd[pi[li - 1 - d]].y == nd[pj[lj - 1 - d]].y;
                    if (!pre) continue;
                    Pair pr{b, j, i, (int)qrow.size(), lj - li};
                    for (int d = lj - li - 1; d >= 0; --d) {      // tokens of A[j] behind A[i], in order
                        qrow.push_back(b);
                        qnode.push_back(nd[pj[d]].x);             // g: the prediction stored at the node before it (-1: root)
                        qtok.push_back(nd[pj[d]].y);
                    }
                    pairs.push_back(pr);
                }
            }
        }
        if (pairs.empty()) return ED_OK;
        const int NQ = (int)qrow.size();
        qout.resize(NQ);
        int32_t* d_row = (int32_t*)(p + w.q_row);
        int32_t* d_node = (int32_t*)(p + w.q_node);
        int32_t* d_tok = (int32_t*)(p + w.q_tok);
        void* d_pred = p + w.q_pred;
        void* d_D1 = p + w.q_D1;
        void* d_hid = p + w.q_hid;
        float* d_logits = (float*)(p + w.q_logits);
        float* d_out = (float*)(p + w.q_out);
        for (int q0 = 0; q0 < NQ; q0 += BEAM_QMAX) {
            const int nq = std::min(BEAM_QMAX, NQ - q0);
            int rc;
            ED_CHECK_HIP(hipMemcpyAsync(d_row, qrow.data() + q0, (size_t)nq * 4, hipMemcpyHostToDevice, s));
            ED_CHECK_HIP(hipMemcpyAsync(d_node, qnode.data() + q0, (size_t)nq * 4, hipMemcpyHostToDevice, s));
            ED_CHECK_HIP(hipMemcpyAsync(d_tok, qtok.data() + q0, (size_t)nq * 4, hipMemcpyHostToDevice, s));
            if (dtype == ED_F32)
                hipLaunchKernelGGL(beam_query_gather<float>, dim3(nq), dim3(64), 0, s, (const float*)node_pred, d_row,
                                   d_node, (float*)d_pred, NODES, P2);
            else
                hipLaunchKernelGGL(beam_query_gather<bf16_t>, dim3(nq), dim3(64), 0, s, (const bf16_t*)node_pred, d_row,
                                   d_node, (bf16_t*)d_pred, NODES, P2);
            if ((rc = edgedict_gemm(dtype, dtype, d_pred, P2, 1, W1d, ldw1, 1, d_D1, J, nq, J, P2, b1, nullptr, 0, 1, s)))
                return rc;
            if (dtype == ED_F32)
                hipLaunchKernelGGL(beam_query_hidden<float>, dim3(nq), dim3(128), 0, s, (const float*)e1t, e_row_stride,
                                   d_row, (const float*)d_D1, (float*)d_hid, J);
            else
                hipLaunchKernelGGL(beam_query_hidden<bf16_t>, dim3(nq), dim3(128), 0, s, (const bf16_t*)e1t, e_row_stride,
                                   d_row, (const bf16_t*)d_D1, (bf16_t*)d_hid, J);
            if ((rc = edgedict_gemm(dtype, ED_F32, d_hid, J, 1, W2, J, 1, d_logits, V, nq, V, J, b2, nullptr, 0, 1, s)))
                return rc;
            hipLaunchKernelGGL(beam_query_logp, dim3(nq), dim3(256), 0, s, d_logits, d_tok, d_out, V);
            ED_CHECK_HIP(hipMemcpyAsync(qout.data() + q0, d_out, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
            ED_CHECK_HIP(hipStreamSynchronize(s));
        }
        // A[j].logp = log_aplusb(A[j].logp, A[i].logp + sum): j ascending, i ascending; A[i] (i > j) still holds the
        // value it had when the frame began (its own turn as a "j" comes later)
        for (const Pair& pr : pairs) {
            double cur = logp_h[(size_t)pr.b * W + pr.i];
            for (int k = 0; k < pr.nq; ++k) cur += (double)qout[pr.q0 + k];
            double& a = logp_h[(size_t)pr.b * W + pr.j];
            const double hi = a > cur ? a : cur;
            a = hi + log1p(exp(-fabs(a - cur)));
            ++prefix_steps;
        }
        ED_CHECK_HIP(hipMemcpyAsync(q.bp_logp, logp_h.data(), logp_h.size() * 8, hipMemcpyHostToDevice, s));
        ED_CHECK_HIP(hipStreamSynchronize(s));       // (logp_h is reused by the next frame)
        return ED_OK;
    };

    const bool fused_step = ed_decode_fused_ok(dtype, emb_dtype, J, V, E, H, P2);
    int cur = 0;
    for (int t = 0; t < maxlen; ++t) {
        const char* e1t = (const char*)E1 + (size_t)t * e_frame_stride * esz;
        if (prefix && t > 0) {
            const int rc = prefix_merge(t, e1t);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(beam_frame_begin, dim3(B), dim3(64), 0, s, q, t, W, EM, V);
        for (int it = 0;; ++it) {
            int rc;
            hipLaunchKernelGGL(beam_pop, dim3(B), dim3(256), 0, s, q, cur, W, V, EM, L, H, B, NODES,
                               bos, pred, h_state, c_state);
            if (fused_step) {
                // prediction-network step + projection + joint hidden + logits as 3 + L fused launches (decode_fused.hip)
                if ((rc = ed_decode_fused_beam_step(dtype, e1t, e_row_stride, B, J, W1d, ldw1, b1, P2, W2, b2, V, emb,
                                                    emb_dtype, E, L, w_ih, w_hh, b_ih, b_hh, H, Wp, bp, h_state, c_state,
                                                    pred, dec_new, hid, logits, h_new, c_new, Y[0], Y[1], s)))
                    return rc;
                if (prefix) {
                    if (dtype == ED_F32)
                        hipLaunchKernelGGL(beam_store_pred<float>, dim3(B), dim3(64), 0, s, q, (const float*)dec_new,
                                           (float*)node_pred, EM, NODES, P2);
                    else
                        hipLaunchKernelGGL(beam_store_pred<bf16_t>, dim3(B), dim3(64), 0, s, q, (const bf16_t*)dec_new,
                                           (bf16_t*)node_pred, EM, NODES, P2);
                }
                hipLaunchKernelGGL(beam_expand, dim3(B), dim3(256), 0, s, q, logits, h_new, c_new, W, V, EM, L, H, B,
                                   blank);
            } else {
            // prediction-network step on y*'s last token from y*'s state (models.py:164,126-132)
            if ((rc = edgedict_embedding_fwd(dtype, emb_dtype, pred, 1, emb, x, B, 1, E, V, 0, 0, s)))
                return rc;
            const void* xin = x;
            int xin_dim = E;
            for (int k = 0; k < L; ++k) {
                if ((rc = edgedict_gemm(dtype, dtype, xin, xin_dim, 1, w_ih[k], xin_dim, 1, G, 4 * H,
                                        B, 4 * H, xin_dim, b_ih[k], b_hh[k], 0, 1, s)))
                    return rc;
                if ((rc = edgedict_lstm_forward(dtype, G, Hprev, Y[k & 1], Cst, w_hh[k], nullptr,
                                                h_state + (size_t)k * B * H, c_state + (size_t)k * B * H,
                                                h_new + (size_t)k * B * H, c_new + (size_t)k * B * H,
                                                B, 1, H, nullptr, s)))
                    return rc;
                xin = Y[k & 1];
                xin_dim = H;
            }
            if ((rc = edgedict_gemm(dtype, dtype, xin, H, 1, Wp, H, 1, dec_new, P2, B, P2, H, bp,
                                    nullptr, 0, 1, s)))
                return rc;
            if (prefix) {        // Sequence.g of the children (models.py:183): this expansion's prediction, kept per node
                if (dtype == ED_F32)
                    hipLaunchKernelGGL(beam_store_pred<float>, dim3(B), dim3(64), 0, s, q, (const float*)dec_new,
                                       (float*)node_pred, EM, NODES, P2);
                else
                    hipLaunchKernelGGL(beam_store_pred<bf16_t>, dim3(B), dim3(64), 0, s, q, (const bf16_t*)dec_new,
                                       (bf16_t*)node_pred, EM, NODES, P2);
            }
            // joint(x_t, pred) (models.py:165): D1 = pred W1d^T + b1; hid = tanh(E1[:,t] + D1); logits
            if ((rc = edgedict_gemm(dtype, dtype, dec_new, P2, 1, W1d, ldw1, 1, D1, J, B, J, P2, b1,
                                    nullptr, 0, 1, s)))
                return rc;
            if (dtype == ED_F32)
                hipLaunchKernelGGL(add_tanh_rows<float>, dim3(ed_grid_for((long long)B * J, 256)),
                                   dim3(256), 0, s, (const float*)e1t, e_row_stride,
                                   (const float*)D1, (float*)hid, B, J);
            else
                hipLaunchKernelGGL(add_tanh_rows<bf16_t>, dim3(ed_grid_for((long long)B * J, 256)),
                                   dim3(256), 0, s, (const bf16_t*)e1t, e_row_stride,
                                   (const bf16_t*)D1, (bf16_t*)hid, B, J);
            if ((rc = edgedict_gemm(dtype, ED_F32, hid, J, 1, W2, J, 1, logits, V, B, V, J, b2,
                                    nullptr, 0, 1, s)))
                return rc;
            hipLaunchKernelGGL(beam_expand, dim3(B), dim3(256), 0, s, q, logits, h_new, c_new, W, V,
                               EM, L, H, B, blank);
            }
            if (it + 1 >= W) {   // B cannot hold W hypotheses before W expansions
                ED_CHECK_HIP(hipMemcpyAsync(open_h.data(), q.open, (size_t)B * 4, hipMemcpyDeviceToHost, s));
                ED_CHECK_HIP(hipStreamSynchronize(s));
                int any = 0;
                for (int b = 0; b < B; ++b) any |= open_h[b];
                if (!any) break;
            }
        }
        hipLaunchKernelGGL(beam_frame_end, dim3(B), dim3(256), 0, s, q, t, cur, W, EM, L, H);
        cur ^= 1;
    }
    ED_CHECK_LAUNCH("beam_search");
    // results: B[0] of the last frame (models.py:201-202), token tree walked on the host
    std::vector<double> logp((size_t)B * W);
    std::vector<int32_t> node((size_t)B * W), nn(B);
    std::vector<int2> nodes((size_t)B * NODES);
    int32_t flags[2];
    long long total = 0;
    ED_CHECK_HIP(hipMemcpyAsync(logp.data(), q.bp_logp, logp.size() * 8, hipMemcpyDeviceToHost, s));
    ED_CHECK_HIP(hipMemcpyAsync(node.data(), q.bp_node, node.size() * 4, hipMemcpyDeviceToHost, s));
    ED_CHECK_HIP(hipMemcpyAsync(nn.data(), q.n_nodes, nn.size() * 4, hipMemcpyDeviceToHost, s));
    ED_CHECK_HIP(hipMemcpyAsync(nodes.data(), q.nodes, nodes.size() * 8, hipMemcpyDeviceToHost, s));
    ED_CHECK_HIP(hipMemcpyAsync(flags, q.flags, 8, hipMemcpyDeviceToHost, s));
    ED_CHECK_HIP(hipMemcpyAsync(&total, q.total_exp, 8, hipMemcpyDeviceToHost, s));
    ED_CHECK_HIP(hipStreamSynchronize(s));
    ED_CHECK_ARG(!flags[0], "beam_search: an utterance needed more than max_expansions = %d expansions in one frame", EM);
    ED_CHECK_ARG(!flags[1], "beam_search: token tree overflow");
    if (expansions_host) *expansions_host = total + prefix_steps;
    std::vector<int32_t> rev;
    for (int b = 0; b < B; ++b) {
        rev.clear();
        for (int n = node[(size_t)b * W]; n >= 0; n = nodes[(size_t)b * NODES + n].x)
            rev.push_back(nodes[(size_t)b * NODES + n].y);
        const int len = (int)rev.size();
        ntokens_host[b] = len;
        ED_CHECK_ARG(len <= max_tokens, "beam_search: hypothesis of %d tokens exceeds max_tokens = %d", len, max_tokens);
        for (int i = 0; i < len; ++i) tokens_host[(size_t)b * max_tokens + i] = rev[len - 1 - i];
        score_host[b] = -logp[(size_t)b * W];
    }
    return ED_OK;
}
