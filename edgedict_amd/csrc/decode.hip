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
                                                   int t, float* __restrict__ score) {
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

    for (int t = 0; t < T; ++t) {
        int rc;
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
                           tokens_out, tok_stride, t, score);
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
