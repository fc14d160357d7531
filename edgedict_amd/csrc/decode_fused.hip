// Fused per-frame kernels of the greedy / streaming / beam RNN-T search (bf16 and fp32 modes), gfx950.
//
// One search frame (Transducer.greedy_decode rnnt/models.py:254-263, PytorchStreamDecoder.decode
// rnnt/stream.py:102-119) is a chain of small products over B rows - joint hidden [B x J x P2], logits
// [B x V x J], the prediction network's L LSTM steps, its projection - each followed by a few elementwise ops.
// Composed from the general kernels that is 11 launches per frame (decode.hip: product, add + tanh, product, pick,
// embedding, L x (product + cell), product, commit), ~7 us each in bf16 and 15-70 us each in fp32 (the general GEMM
// on 2-16 tiles), with nothing else to overlap them.  Here a frame is FIVE launches, each a product with its whole
// epilogue:
//   1. dec_joint_hidden   hid = tanh(E1[:, t] + dec_out W1d^T + b1)
//   2. dec_logits_pick    logits = hid W2^T + b2 in registers -> per row and 64-column slice: arg-max, arg-max
//                         without <unk>, sum of exponentials (the logits never reach memory)
//   3. dec_lstm_step      (first layer) final pick from the slices (+ the stream decoder's <unk> rule, + score),
//                         token out, embedding, x W_ih^T + h W_hh^T + b, LSTM cell
//   4. dec_lstm_step      (further layers)
//   5. dec_proj_commit    dec_new = h W_p^T + b_p; where the symbol is not blank: dec_out, h, c <- new
// Operands go straight from L2 into MFMA fragments, the weights as the FIRST operand so that a lane ends up with 4
// consecutive output columns of one row; no LDS staging - every operand is used once per workgroup.  These kernels
// are LATENCY-bound (a wave or four per CU, nothing else to hide an L2 round trip behind), so the rule is: one wave =
// ONE 16 x 16 output tile, every operand fragment of its K loop requested at once (up to 20 k-steps = 160 registers),
// a second round trip only where the data dependence forces one (the embedding row of the symbol just picked).
//   bf16: v_mfma_f32_16x16x32_bf16, a lane's 16-byte load = 8 k of one k-step;
//   fp32 (the parity mode, pinned token-exact on the reference's goldens): v_mfma_f32_16x16x4_f32, a lane's 16-byte load
//         = 4 k that feed FOUR MFMA steps (the same k permutation on both operands - see lstm.hip f32_product16).
// Rows are independent and a row's arithmetic does not depend on its position in the batch, so S concurrent streams
// equal S single-stream decoders bit for bit (tests/test_stream_gpu.py).
#include "common.hpp"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ bf16x8_t ld8(const bf16_t* p) { return *reinterpret_cast<const bf16x8_t*>(p); }
__device__ __forceinline__ bf16x8_t ld8_f32(const float* p) {       // 8 fp32 -> bf16 fragment
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    union { unsigned u[4]; bf16x8_t v; } r;
    r.u[0] = f32x2_to_bf16x2(a.x, a.y); r.u[1] = f32x2_to_bf16x2(a.z, a.w);
    r.u[2] = f32x2_to_bf16x2(b.x, b.y); r.u[3] = f32x2_to_bf16x2(b.z, b.w);
    return r.v;
}
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) {
    const float e = __expf(-2.f * fabsf(x));
    const float t = (1.f - e) / (1.f + e);
    return x < 0.f ? -t : t;
}

// ---- the two arithmetic modes.  ET = element type of the weights and of the activations between the kernels.
// KS = k per "k-step" (one 16-byte load per lane and operand), LK = k per lane.
template <typename ET> struct Mode;
template <> struct Mode<bf16_t> {
    static constexpr int KS = 32, LK = 8;
    typedef bf16x8_t frag;
    static __device__ __forceinline__ frag ldw(const bf16_t* p) { return ld8(p); }
    static __device__ __forceinline__ frag ldx(const bf16_t* p) { return ld8(p); }
    static __device__ __forceinline__ frag ldx(const float* p) { return ld8_f32(p); }
    static __device__ __forceinline__ void mma(f32x4_t& acc, const frag& w, const frag& x) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ float act(float v) { return bf16_to_f32(f32_to_bf16(v)); }   // value as stored
    static __device__ __forceinline__ float ld1(const bf16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void st4(bf16_t* p, float a, float b, float c, float d) {
        uint2 o;
        o.x = f32x2_to_bf16x2(a, b);
        o.y = f32x2_to_bf16x2(c, d);
        *reinterpret_cast<uint2*>(p) = o;
    }
    static __device__ __forceinline__ void ld4(const bf16_t* p, float (&v)[4]) {
        const uint2 e = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(e.x << 16); v[1] = __uint_as_float(e.x & 0xffff0000u);
        v[2] = __uint_as_float(e.y << 16); v[3] = __uint_as_float(e.y & 0xffff0000u);
    }
    static __device__ __forceinline__ void st1(bf16_t* p, float v) { *p = f32_to_bf16(v); }
    static __device__ __forceinline__ float sig(float x) { return sigm(x); }
    static __device__ __forceinline__ float tnh(float x) { return tanh_fast(x); }
};
template <> struct Mode<float> {
    static constexpr int KS = 16, LK = 4;
    typedef float4 frag;
    static __device__ __forceinline__ frag ldw(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ frag ldx(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ void mma(f32x4_t& acc, const frag& w, const frag& x) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, x.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, x.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, x.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, x.w, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ float act(float v) { return v; }
    static __device__ __forceinline__ float ld1(const float* p) { return *p; }
    static __device__ __forceinline__ void st4(float* p, float a, float b, float c, float d) {
        *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
    }
    static __device__ __forceinline__ void ld4(const float* p, float (&v)[4]) {
        const float4 e = *reinterpret_cast<const float4*>(p);
        v[0] = e.x; v[1] = e.y; v[2] = e.z; v[3] = e.w;
    }
    static __device__ __forceinline__ void st1(float* p, float v) { *p = v; }
    // the parity mode keeps the exact functions of the composed path (lstm.hip lstm_step_fwd<float>)
    static __device__ __forceinline__ float sig(float x) { return 1.f / (1.f + expf(-x)); }
    static __device__ __forceinline__ float tnh(float x) { return tanhf(x); }
};

// acc (+)= W[wrow][k] x X[xrow][k] over K (a multiple of KS): lane (r16 = lane & 15, q = lane >> 4) ends with
// acc[e] = out[xrow of lane r16][column of W row 4 q + e].  wp / xp point at this lane's W row / X row (+ q * LK).
// CH k-steps are requested together, then multiplied, k ascending (the order of the sums does not depend on CH).
template <typename ET, int CH, typename XT>
__device__ __forceinline__ void wave_product(f32x4_t& acc, const XT* __restrict__ xp, const ET* __restrict__ wp, int K) {
    typedef Mode<ET> Md;
    for (int k0 = 0; k0 < K; k0 += Md::KS * CH) {
        typename Md::frag a[CH], b[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int k = min(k0 + Md::KS * c, K - Md::KS);      // (clamped: a repeated k-step is skipped below)
            b[c] = Md::ldw(wp + k);
            a[c] = Md::ldx(xp + k);
        }
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (k0 + Md::KS * c < K) Md::mma(acc, b[c], a[c]);
    }
}

// ---------------------------------------------------------------------------------------------- 1
// hid[b, j] = tanh(E1[b, t, j] + act(dec_out[b] . W1d[j] + b1[j]));  one wave per (16 rows, 16 columns), 4 waves
template <typename ET>
__global__ __launch_bounds__(256) void dec_joint_hidden(const ET* __restrict__ E1t, long long e_row_stride,
                                                        const ET* __restrict__ dec_out, int P2,
                                                        const ET* __restrict__ W1d, long long ldw1,
                                                        const float* __restrict__ b1, ET* __restrict__ hid,
                                                        int B, int J) {
    typedef Mode<ET> Md;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, q = lane >> 4;
    const int row = blockIdx.x * 16 + r16, col0 = blockIdx.y * 64 + wave * 16;
    if (col0 >= J) return;
    // (everything the epilogue needs is requested BEFORE the product: behind it, it would be one more L2 round trip)
    const int c = min(col0 + q * 4, J - 4);       // (J % 4 == 0: a lane's 4 columns are in or out together)
    float ev[4], h[4];
    Md::ld4(E1t + (long long)min(row, B - 1) * e_row_stride + c, ev);
    const float4 bv = *reinterpret_cast<const float4*>(b1 + c);
    f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    wave_product<ET, 20>(acc, dec_out + (long long)min(row, B - 1) * P2 + q * Md::LK,
                         W1d + (long long)min(col0 + r16, J - 1) * ldw1 + q * Md::LK, P2);
    if (row >= B || col0 + q * 4 >= J) return;
    const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = tanhf(ev[i] + Md::act(acc[i] + bb[i]));
    Md::st4(hid + (long long)row * J + c, h[0], h[1], h[2], h[3]);
}

// ---------------------------------------------------------------------------------------------- 2
// per row and 64-column slice of the logits: (max, first arg-max), the same without column `unk`, sum of
// exp(z - max).  4 waves = 4 x 16 columns of 16 rows.  With logits_out the logits are written instead (fp32 [B, V]:
// the beam search expands every symbol).
struct PickPart {
    float m; int a;          // max over the slice, lowest index on ties
    float mx; int ax;        // ... over the slice without column unk (-inf / INT_MAX if nothing is left)
    float se;                // sum exp(z - m)
    int pad[3];
};
__device__ __forceinline__ void pick_merge(float& m, int& a, float om, int oa) {
    if (om > m || (om == m && oa < a)) { m = om; a = oa; }
}
template <typename ET>
__global__ __launch_bounds__(256) void dec_logits_pick(const ET* __restrict__ hid, int J, const ET* __restrict__ W2,
                                                       const float* __restrict__ b2, int V, int unk, int want_sum,
                                                       PickPart* __restrict__ parts, float* __restrict__ logits_out,
                                                       int B) {
    typedef Mode<ET> Md;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, q = lane >> 4;
    const int row = blockIdx.x * 16 + r16, col0 = blockIdx.y * 64 + wave * 16;
    const float4 bv = *reinterpret_cast<const float4*>(b2 + min(col0 + q * 4, V - 4));     // (V % 4 == 0), before the product
    f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    wave_product<ET, 20>(acc, hid + (long long)min(row, B - 1) * J + q * Md::LK,
                         W2 + (long long)min(col0 + r16, V - 1) * J + q * Md::LK, J);
    const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
    float z[4];
    float m = -INFINITY, mx = -INFINITY;
    int a = 0x7fffffff, ax = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = col0 + q * 4 + i;
        const float v = c < V ? acc[i] + bb[i] : -INFINITY;
        z[i] = v;
        if (v > m) { m = v; a = c; }                       // ascending c within a lane: strict > keeps the first
        if (c != unk && v > mx) { mx = v; ax = c; }
    }
    if (logits_out) {
        const int c = col0 + q * 4;
        if (row < B && c < V) {                            // (V % 4 == 0)
            *reinterpret_cast<float4*>(logits_out + (long long)row * V + c) = make_float4(z[0], z[1], z[2], z[3]);
        }
        return;
    }
    // the 4 lanes that hold this row (q = 0..3), then the 4 waves
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
        pick_merge(m, a, __shfl_xor(m, off, 64), __shfl_xor(a, off, 64));
        pick_merge(mx, ax, __shfl_xor(mx, off, 64), __shfl_xor(ax, off, 64));
    }
    __shared__ float s_m[4][16], s_mx[4][16], s_se[4][16];
    __shared__ int s_a[4][16], s_ax[4][16];
    if (q == 0) { s_m[wave][r16] = m; s_a[wave][r16] = a; s_mx[wave][r16] = mx; s_ax[wave][r16] = ax; }
    __syncthreads();
    float gm = s_m[0][r16];
#pragma unroll
    for (int w = 1; w < 4; ++w) gm = fmaxf(gm, s_m[w][r16]);
    if (want_sum) {
        float se = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) se += __expf(z[i] - gm);     // (-inf columns add 0; gm is finite: V > 0)
        se += __shfl_xor(se, 16, 64);
        se += __shfl_xor(se, 32, 64);
        if (q == 0) s_se[wave][r16] = se;
        __syncthreads();
    }
    if (threadIdx.x < 16 && row < B) {
        PickPart p;
        p.m = s_m[0][r16]; p.a = s_a[0][r16]; p.mx = s_mx[0][r16]; p.ax = s_ax[0][r16];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            pick_merge(p.m, p.a, s_m[w][r16], s_a[w][r16]);
            pick_merge(p.mx, p.ax, s_mx[w][r16], s_ax[w][r16]);
        }
        p.se = want_sum ? s_se[0][r16] + s_se[1][r16] + s_se[2][r16] + s_se[3][r16] : 0.f;
        p.pad[0] = p.pad[1] = p.pad[2] = 0;
        parts[(long long)row * gridDim.y + blockIdx.y] = p;
    }
}

// ---------------------------------------------------------------------------------------------- 3 / 4
// One LSTM step of one layer for 16 rows x 16 units per workgroup (wave g = gate g).  first != 0: the layer input is
// the embedding of the symbol picked from `parts` (which this kernel finishes: every workgroup for its own rows, the
// unit-block-0 workgroups write it out); otherwise x = y_prev (rows of the layer below).
// Everything that does not depend on the pick - W_ih, W_hh, h - is requested BEFORE the slices are merged (16 lanes per
// row, each a share of the slices, a butterfly over the 16), the embedding rows right behind it: two L2 round trips
// per launch where the serial merge + two products made 3 + nslices.
template <typename ET>
struct LstmStepArgs {
    const PickPart* parts; int nslices; int unk; int blank;
    int32_t* pred; int32_t* tokens; long long tok_stride; int t; float* score;
    const void* emb; int emb_f32; int E;
    const ET* x_prev;                      // [B, H] (layers > 0)
    const ET* w_ih; const ET* w_hh; const float* b_ih; const float* b_hh;
    const float* h_in; const float* c_in;  // [B, H] fp32 state of this layer
    float* h_out; float* c_out;            // [B, H] fp32 candidates
    ET* y_out;                             // [B, H] copy of h_out in the mode's type (next layer / projection operand)
    int B, H, Kx;
    int V;                                 // rows of the embedding table: every symbol that indexes it is clamped to [0, V)
};
template <typename ET>
__global__ __launch_bounds__(256) void dec_lstm_step(LstmStepArgs<ET> A, int first) {
    typedef Mode<ET> Md;
    constexpr int CH = 8;
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6, r16 = lane & 15, q = lane >> 4;
    const int row0 = blockIdx.x * 16, j0 = blockIdx.y * 16;
    const int row = row0 + r16, rowc = min(row, A.B - 1);
    __shared__ int s_pred[16];
    __shared__ float s_gate[4][16][17];
    const int col0 = g * A.H + j0;                 // W rows of this gate's 16 units
    const ET* wip = A.w_ih + (long long)(col0 + r16) * A.Kx + q * Md::LK;
    const ET* whp = A.w_hh + (long long)(col0 + r16) * A.H + q * Md::LK;
    const float* hp = A.h_in + (long long)rowc * A.H + q * Md::LK;
    // ---- the epilogue's operands first (behind the products they would be one more L2 round trip each) ...
    const float4 bi4 = *reinterpret_cast<const float4*>(A.b_ih + col0 + q * 4), bh4 = *reinterpret_cast<const float4*>(A.b_hh + col0 + q * 4);
    const int cr = threadIdx.x & 15, cu = threadIdx.x >> 4;          // cells: thread <-> (row tid & 15, unit tid >> 4)
    const int crow = row0 + cr;
    const long long co = (long long)min(crow, A.B - 1) * A.H + j0 + cu;
    const float c_prev = A.c_in[co];
    // ---- ... then batch 0 (the first CH k-steps) of both products
    typename Md::frag wi[CH], wh[CH], hx[CH], xx[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        wi[c] = Md::ldw(wip + min(c * Md::KS, A.Kx - Md::KS));
        wh[c] = Md::ldw(whp + min(c * Md::KS, A.H - Md::KS));
        hx[c] = Md::ldx(hp + min(c * Md::KS, A.H - Md::KS));
    }
    const ET* xp = nullptr;                        // this lane's x row (layers > 0) ...
    const float* xpf = nullptr;                    // ... or embedding row (fp32 table)
    if (!first) {
        xp = A.x_prev + (long long)rowc * A.H + q * Md::LK;
#pragma unroll
        for (int c = 0; c < CH; ++c) xx[c] = Md::ldx(xp + min(c * Md::KS, A.Kx - Md::KS));
    } else {
        if (first == 2) {                          // beam search: the symbol is the popped hypothesis' last token
            if (threadIdx.x < 16) s_pred[threadIdx.x] = min(max(A.pred[min(row0 + (int)threadIdx.x, A.B - 1)], 0), A.V - 1);
        } else {
            // thread <-> (row tid >> 4, slice share tid & 15)
            const int pr = threadIdx.x >> 4, ps = threadIdx.x & 15;
            const int rr = min(row0 + pr, A.B - 1);
            const PickPart* p = A.parts + (long long)rr * A.nslices;
            float m = -INFINITY, mx = -INFINITY;
            int a = 0x7fffffff, ax = 0x7fffffff;
            for (int s = ps; s < A.nslices; s += 16) {
                pick_merge(m, a, p[s].m, p[s].a);
                pick_merge(mx, ax, p[s].mx, p[s].ax);
            }
#pragma unroll
            for (int off = 1; off <= 8; off <<= 1) {
                pick_merge(m, a, __shfl_xor(m, off, 64), __shfl_xor(a, off, 64));
                pick_merge(mx, ax, __shfl_xor(mx, off, 64), __shfl_xor(ax, off, 64));
            }
            int pick = a;
            // rnnt/stream.py:105-108: an arg-max that is <unk> has its logit set to 0 and the arg-max is retaken
            if (A.unk >= 0 && a == A.unk) pick = (0.f > mx || (0.f == mx && A.unk < ax)) ? A.unk : ax;
            // a row without ONE comparable logit (all NaN, or all -inf: `v > m` never held) leaves the sentinel: such a
            // stream emits blank - its prediction network does not advance, nothing is read outside the embedding table,
            // and the other rows of the batch are untouched (the composed path clamped in edgedict_embedding_fwd)
            if ((unsigned)pick >= (unsigned)A.V) pick = min(max(A.blank, 0), A.V - 1);
            if (ps == 0) s_pred[pr] = pick;
            if (blockIdx.y == 0 && row0 + pr < A.B) {
                if (ps == 0) {
                    A.pred[rr] = pick;
                    if (A.tokens) A.tokens[(long long)rr * A.tok_stride + A.t] = pick;
                }
                if (A.score) {                     // -(max log p) = log sum exp(z - max)   (greedy mode: unk < 0)
                    float se = 0.f;
                    for (int s = ps; s < A.nslices; s += 16) se += p[s].se * __expf(p[s].m - m);
#pragma unroll
                    for (int off = 1; off <= 8; off <<= 1) se += __shfl_xor(se, off, 64);
                    if (ps == 0) A.score[rr] += logf(se);
                }
            }
        }
        __syncthreads();
        const int tok = s_pred[r16];
        if (A.emb_f32) {
            xpf = reinterpret_cast<const float*>(A.emb) + (long long)tok * A.E + q * Md::LK;
#pragma unroll
            for (int c = 0; c < CH; ++c) xx[c] = Md::ldx(xpf + min(c * Md::KS, A.Kx - Md::KS));
        } else {
            if constexpr (sizeof(ET) == 2) {
                xp = reinterpret_cast<const ET*>(A.emb) + (long long)tok * A.E + q * Md::LK;
#pragma unroll
                for (int c = 0; c < CH; ++c) xx[c] = Md::ldx(xp + min(c * Md::KS, A.Kx - Md::KS));
            }
        }
    }
    // ---- x W_ih^T, then h W_hh^T, k ascending
    f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < CH; ++c)
        if (c * Md::KS < A.Kx) Md::mma(acc, wi[c], xx[c]);
    if (A.Kx > CH * Md::KS) {
        const int k0 = CH * Md::KS;
        if (xpf) wave_product<ET, CH>(acc, xpf + k0, wip + k0, A.Kx - k0);
        else wave_product<ET, CH>(acc, xp + k0, wip + k0, A.Kx - k0);
    }
#pragma unroll
    for (int c = 0; c < CH; ++c)
        if (c * Md::KS < A.H) Md::mma(acc, wh[c], hx[c]);
    if (A.H > CH * Md::KS) wave_product<ET, CH>(acc, hp + CH * Md::KS, whp + CH * Md::KS, A.H - CH * Md::KS);
    {
        const float bi[4] = {bi4.x, bi4.y, bi4.z, bi4.w}, bh[4] = {bh4.x, bh4.y, bh4.z, bh4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) s_gate[g][r16][q * 4 + i] = acc[i] + bi[i] + bh[i];
    }
    __syncthreads();
    if (crow < A.B) {
        const float ig = Md::sig(s_gate[0][cr][cu]), fg = Md::sig(s_gate[1][cr][cu]);
        const float gg = Md::tnh(s_gate[2][cr][cu]), og = Md::sig(s_gate[3][cr][cu]);
        const long long o = co;
        const float c = fg * c_prev + ig * gg;
        const float h = og * Md::tnh(c);
        A.c_out[o] = c;
        A.h_out[o] = h;
        Md::st1(A.y_out + o, h);
    }
}

// ---------------------------------------------------------------------------------------------- 5
// dec_new = y W_p^T + b_p for 16 rows x 64 columns per workgroup (a wave = 16 columns); rows whose symbol is not blank
// take dec_new and the new prediction-network state (each block moves its share of the L x H state values of its rows)
template <typename ET>
__global__ __launch_bounds__(256) void dec_proj_commit(const ET* __restrict__ y, int H, const ET* __restrict__ Wp,
                                                       const float* __restrict__ bp, int P2,
                                                       const int32_t* __restrict__ pred, int blank,
                                                       ET* __restrict__ dec_out, float* __restrict__ h_state,
                                                       const float* __restrict__ h_new, float* __restrict__ c_state,
                                                       const float* __restrict__ c_new, int L, int B) {
    typedef Mode<ET> Md;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, q = lane >> 4;
    const int row0 = blockIdx.x * 16, row = row0 + r16, col0 = blockIdx.y * 64 + wave * 16;
    __shared__ int s_keep[16];
    if (threadIdx.x < 16) {
        const int r = row0 + (int)threadIdx.x;
        s_keep[threadIdx.x] = r < B && (pred == nullptr || pred[r] != blank);          // (pred == null: every row)
    }
    const int c = col0 + q * 4;
    const float4 bv = *reinterpret_cast<const float4*>(bp + min(c, P2 - 4));       // (P2 % 4 == 0), before the product
    f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (col0 < P2)
        wave_product<ET, 16>(acc, y + (long long)min(row, B - 1) * H + q * Md::LK,
                             Wp + (long long)min(col0 + r16, P2 - 1) * H + q * Md::LK, H);
    __syncthreads();
    if (s_keep[r16] && c < P2)
        Md::st4(dec_out + (long long)row * P2 + c, acc[0] + bv.x, acc[1] + bv.y, acc[2] + bv.z, acc[3] + bv.w);
    if (pred == nullptr) return;
    // state commit: this block's slice of [L][16 rows][H], one value per thread and pass
    const int per = (L * H + gridDim.y - 1) / gridDim.y;
    const int lo = blockIdx.y * per, n = min(L * H, lo + per) - lo;
    for (int idx = threadIdx.x; idx < 16 * n; idx += 256) {
        const int rr = idx / n, i = lo + idx - rr * n;
        if (!s_keep[rr]) continue;
        const int l = i / H, j = i - l * H;
        const long long o = ((long long)l * B + row0 + rr) * H + j;
        h_state[o] = h_new[o];
        c_state[o] = c_new[o];
    }
}

// ---------------------------------------------------------------------------------------------- stream encoder
// One LSTM time step of one encoder layer for 16 RT rows x 16 units per workgroup: the streaming decoder's encoder
// advances S streams by a frame or two per chunk, and composed from the per-layer kernels (LayerNorm, input product,
// state copy, kernel-per-step recurrence) that was 4 launches and ~35 us per layer-frame plus their host glue.
// 8 waves: wave (g, half) = gate g of the 16 units, half 0 the input product x W_ih^T, half 1 the recurrent product
// h W_hh^T (h as bf16: the previous step's y row, or the bf16 copy of the carried state made once per call - the same
// values the conversion at load time would produce) - two independent K loops side by side, 8 k-steps of both operands
// requested at once (one L2 round trip per 256 k), the W fragment of a k-step used for all RT row tiles.
// Loads are never conditional (a predicated load is its own exec-masked block and the compiler waits for each - the first
// version of this kernel spent 48 of its 60 us at S = 256 in 32 exposed round trips): addresses are clamped into the
// row and the W fragment of a k-group past K is zeroed after the load (the encoder's first layer has K = 240).
template <int RT, int CH>
__device__ __forceinline__ void rows_product(f32x4_t (&acc)[RT], const bf16_t* const (&xp)[RT], const bf16_t* wp, int K,
                                             int kq) {
    const int KP = (K + 31) & ~31;
    for (int k0 = 0; k0 < KP; k0 += 32 * CH) {
        bf16x8_t a[CH][RT], b[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int k = min(k0 + 32 * c, KP - 32);
            const int kk = min(k + kq * 8, K - 8);           // K % 8 == 0: the last whole 8-group of the row
            b[c] = ld8(wp + kk);
#pragma unroll
            for (int m = 0; m < RT; ++m) a[c][m] = ld8(xp[m] + kk);
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (k0 + 32 * c < KP) {
                union { unsigned u[4]; bf16x8_t v; } z;
                z.v = b[c];
                if (k0 + 32 * c + kq * 8 >= K) z.u[0] = z.u[1] = z.u[2] = z.u[3] = 0u;
#pragma unroll
                for (int m = 0; m < RT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(z.v, a[c][m], acc[m], 0, 0, 0);
            }
        }
    }
}

template <int RT>
__global__ __launch_bounds__(512) void enc_lstm_step(const bf16_t* __restrict__ x, long long ldx, int Kx,
                                                     const bf16_t* __restrict__ w_ih, const bf16_t* __restrict__ w_hh,
                                                     const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                     const bf16_t* __restrict__ hb, long long ldh,
                                                     const float* c_in, float* __restrict__ h_out, float* c_out,
                                                     bf16_t* __restrict__ y, long long ldy,
                                                     int B, int H) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = wave & 3, half = wave >> 2;
    const int r16 = lane & 15, q = lane >> 4;
    const int row0 = blockIdx.x * 16 * RT, j0 = blockIdx.y * 16;
    __shared__ float s_gate[2][4][16 * RT][17];
    f32x4_t acc[RT];
#pragma unroll
    for (int m = 0; m < RT; ++m) acc[m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int col = g * H + j0 + r16;                       // this lane's W row (gate g, unit j0 + r16)
    const bf16_t* xp[RT];
    const bf16_t* src = half ? hb : x;
    const long long ld = half ? ldh : ldx;
    const int K = half ? H : Kx;
#pragma unroll
    for (int m = 0; m < RT; ++m) xp[m] = src + (long long)min(row0 + m * 16 + r16, B - 1) * ld;
    rows_product<RT, RT >= 4 ? 8 : 16>(acc, xp, (half ? w_hh : w_ih) + (long long)col * K, K, q);
#pragma unroll
    for (int m = 0; m < RT; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) s_gate[half][g][m * 16 + r16][q * 4 + i] = acc[m][i];
    __syncthreads();
    for (int cell = threadIdx.x; cell < 256 * RT; cell += 512) {
        const int cr = cell % (16 * RT), cu = cell / (16 * RT);
        const int crow = row0 + cr;
        if (crow >= B) continue;
        float pre[4];
#pragma unroll
        for (int gg = 0; gg < 4; ++gg)
            pre[gg] = (s_gate[0][gg][cr][cu] + s_gate[1][gg][cr][cu]) + b_ih[gg * H + j0 + cu] + b_hh[gg * H + j0 + cu];
        const float ig = sigm(pre[0]), fg = sigm(pre[1]), gt = tanh_fast(pre[2]), og = sigm(pre[3]);
        const long long o = (long long)crow * H + j0 + cu;
        const float c = fg * c_in[o] + ig * gt;
        const float h = og * tanh_fast(c);
        c_out[o] = c;
        h_out[o] = h;
        y[(long long)crow * ldy + j0 + cu] = f32_to_bf16(h);
    }
}

// The same layer-frame for MANY rows (17 .. thousands of streams): a 64-row x (16 units x 4 gates) output tile per
// workgroup with both operands staged through LDS - the one-tile-per-wave form above re-reads the 64-row X / h block
// once per gate wave (1.3 MB per workgroup at S = 256: 42 us per layer-frame at the 49 GB/s a CU pulls), here every
// operand byte enters the CU once (512 KB: ~11 us).  The K loop is gemm_nt_ring64's (gemm_nt.hip): ring of three
// 64-k stages filled by LDS-DMA (global_load_lds_dwordx4, wave w brings pieces w and w + 4 of both operands), counted
// `s_waitcnt vmcnt(4)` + raw barrier, XOR-swizzled 16-byte chunks; it runs over [x | h] x [W_ih | W_hh]^T as two K
// segments with SEPARATE accumulators, summed as (x part + h part) + b_ih + b_hh - bit for bit what enc_lstm_step<1>
// computes for the same row, so a stream's state does not depend on how many streams share its batch.  K tails (the
// first layer's K = 240): chunks past K are fetched from a 16-byte zero block.
__device__ const uint4 ed_zero16 = {0u, 0u, 0u, 0u};
__device__ __forceinline__ void glds16(const void* src, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__global__ __launch_bounds__(256, 2) void enc_lstm_tile(const bf16_t* __restrict__ x, long long ldx, int Kx,
                                                        const bf16_t* __restrict__ w_ih, const bf16_t* __restrict__ w_hh,
                                                        const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                        const bf16_t* __restrict__ hb, long long ldh, const float* c_in,
                                                        float* __restrict__ h_out, float* c_out, bf16_t* __restrict__ y,
                                                        long long ldy, int B, int H) {
    constexpr int BKK = 64, NS = 3, A_BYTES = 64 * BKK * 2, BUF_BYTES = 2 * A_BYTES;      // 16 KB per stage
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * BUF_BYTES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * 64, j0 = blockIdx.y * 16;
    const int KT0 = (Kx + BKK - 1) / BKK, KT = KT0 + (H + BKK - 1) / BKK;
    const int prow = lane >> 3;
    const int chunk = (lane & 7) ^ prow;
    const bf16_t* ax[2];
    const bf16_t* ah[2];
    const bf16_t* bx[2];
    const bf16_t* bh[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (i * 4 + wave) * 8 + prow;                       // row of the tile / local gate column
        const long long row = min(m0 + r, B - 1);
        const long long wrow = (long long)(r >> 4) * H + j0 + (r & 15);   // W row of (gate r >> 4, unit j0 + (r & 15))
        ax[i] = x + row * ldx + chunk * 8;
        ah[i] = hb + row * ldh + chunk * 8;
        bx[i] = w_ih + wrow * Kx + chunk * 8;
        bh[i] = w_hh + wrow * H + chunk * 8;
    }
    auto issue = [&](int q) {
        unsigned char* base = smem + (q % NS) * BUF_BYTES;
        const bool seg = q >= KT0;
        const int k0 = (seg ? q - KT0 : q) * BKK;
        const bool in = k0 + chunk * 8 < (seg ? H : Kx);               // (K % 8 == 0: a chunk is in or out as a whole)
#pragma unroll
        for (int i = 0; i < 2; ++i)
            glds16(in ? (const void*)((seg ? ah[i] : ax[i]) + k0) : (const void*)&ed_zero16, base + (i * 4 + wave) * 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            glds16(in ? (const void*)((seg ? bh[i] : bx[i]) + k0) : (const void*)&ed_zero16, base + A_BYTES + (i * 4 + wave) * 1024);
    };
    f32x4_t accx[2][2], acch[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) accx[i][j] = acch[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    auto stage = [&](int q, f32x4_t (&acc)[2][2]) {
        if (q + 1 < KT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");          // stage q landed everywhere; stage q - 1 is read out
        if (q + 2 < KT) issue(q + 2);
        const unsigned char* sA = smem + (q % NS) * BUF_BYTES;
        const unsigned char* sB = sA + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < BKK / 32; ++ks) {
            bf16x8_t a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ra = wm * 32 + i * 16 + r16;
                a[i] = *reinterpret_cast<const bf16x8_t*>(sA + ra * 128 + (((ks * 4 + kq) ^ (ra & 7)) << 4));
                const int rb = wn * 32 + i * 16 + r16;
                b[i] = *reinterpret_cast<const bf16x8_t*>(sB + rb * 128 + (((ks * 4 + kq) ^ (rb & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
    };
    issue(0);
    if (KT > 1) issue(1);
    for (int q = 0; q < KT0; ++q) stage(q, accx);
    for (int q = KT0; q < KT; ++q) stage(q, acch);
    __syncthreads();
    float* sg = reinterpret_cast<float*>(smem);           // [64 rows][65]: x part + h part of every pre-activation
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int nl = wn * 32 + j * 16 + kq * 4, ml = wm * 32 + i * 16 + r16;
#pragma unroll
            for (int e = 0; e < 4; ++e) sg[ml * 65 + nl + e] = accx[i][j][e] + acch[i][j][e];
        }
    __syncthreads();
    for (int cell = threadIdx.x; cell < 64 * 16; cell += 256) {
        const int cr = cell & 63, cu = cell >> 6;
        const int crow = m0 + cr;
        if (crow >= B) continue;
        float pre[4];
#pragma unroll
        for (int gg = 0; gg < 4; ++gg)
            pre[gg] = sg[cr * 65 + gg * 16 + cu] + b_ih[gg * H + j0 + cu] + b_hh[gg * H + j0 + cu];
        const float ig = sigm(pre[0]), fg = sigm(pre[1]), gt = tanh_fast(pre[2]), og = sigm(pre[3]);
        const long long o = (long long)crow * H + j0 + cu;
        const float c = fg * c_in[o] + ig * gt;
        const float h = og * tanh_fast(c);
        c_out[o] = c;
        h_out[o] = h;
        y[(long long)crow * ldy + j0 + cu] = f32_to_bf16(h);
    }
}

inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace

// ---- C ABI: the streaming encoder step (rnnt/stream.py:93-100: `self.encoder(xs, (enc_h, enc_c))` on a chunk of a
// few frames).  Encoder.forward rnnt/models.py:131-136 / ResLayerNormLSTM.forward :55-75 for T <= a few frames, bf16:
// input LayerNorm, then per layer T fused LSTM steps + residual / LayerNorm / TimeReduction; ONE native call.
extern "C" size_t edgedict_stream_encoder_workspace_bytes(int B, int T, int I0, int H, int L) {
    if (B <= 0 || T <= 0 || L <= 0) return 0;
    const size_t D = (size_t)(I0 > H ? I0 : H);
    // X0 (normalised input), two activation buffers, Y, two state scratch buffers, statistics
    return 4 * align256((size_t)B * T * D * 2) + 2 * align256((size_t)B * H * 4) + 2 * align256((size_t)B * T * 4) +
           align256((size_t)L * B * H * 2);      // + the bf16 copy of the carried h
}

extern "C" int edgedict_stream_encoder_step(const void* xs, int x_dtype, int B, int T, int I0, int H, int L,
                                            const float* in_gamma, const float* in_beta,
                                            const void* const* w_ih, const void* const* w_hh,
                                            const float* const* b_ih, const float* const* b_hh,
                                            const float* const* ln_gamma, const float* const* ln_beta,
                                            const int* reduce, float* h_state, float* c_state, void* out,
                                            int* T_out, void* workspace, void* stream_) {
    ED_CHECK_ARG(xs && in_gamma && in_beta && w_ih && w_hh && b_ih && b_hh && ln_gamma && ln_beta && reduce && h_state &&
                     c_state && out && workspace && T_out, "stream_encoder_step: null pointer");
    ED_CHECK_ARG(B > 0 && T > 0 && L > 0 && H % 32 == 0 && I0 % 8 == 0, "stream_encoder_step: need H %% 32 == 0 and "
                 "I0 %% 8 == 0 (B=%d T=%d I0=%d H=%d)", B, T, I0, H);
    ED_CHECK_ARG(x_dtype == ED_F32 || x_dtype == ED_BF16, "stream_encoder_step: bad input dtype");
    hipStream_t s = (hipStream_t)stream_;
    const size_t D = (size_t)(I0 > H ? I0 : H);
    char* p = (char*)workspace;
    const size_t act = align256((size_t)B * T * D * 2);
    bf16_t* X = (bf16_t*)p;                    // layer input  [B, T_l, K_l]
    bf16_t* Xn = (bf16_t*)(p + act);           // next layer's input
    bf16_t* Y = (bf16_t*)(p + 2 * act);        // h rows       [B, T_l, H]
    bf16_t* Xc = (bf16_t*)(p + 3 * act);       // bf16 copy of an fp32 input
    float* mean = (float*)(p + 4 * act + 2 * align256((size_t)B * H * 4));
    float* rstd = (float*)((char*)mean + align256((size_t)B * T * 4));
    bf16_t* HB = (bf16_t*)((char*)rstd + align256((size_t)B * T * 4));
    int rc;
    if ((rc = edgedict_cast(ED_F32, h_state, ED_BF16, HB, (long long)L * B * H, s))) return rc;
    const void* x_in = xs;
    if (x_dtype == ED_F32) {                   // the feature front-end hands over fp32
        if ((rc = edgedict_cast(ED_F32, xs, ED_BF16, Xc, (long long)B * T * I0, s))) return rc;
        x_in = Xc;
    }
    if ((rc = edgedict_layernorm_fwd(ED_BF16, x_in, nullptr, in_gamma, in_beta, X, mean, rstd, B, T, I0, 1, 1e-5f, s)))
        return rc;
    int Tl = T, K = I0;
    for (int l = 0; l < L; ++l) {
        ED_CHECK_ARG(reduce[l] == 1 || reduce[l] == 2, "stream_encoder_step: time reduction must be 1 or 2");
        float* hl = h_state + (size_t)l * B * H;
        float* cl = c_state + (size_t)l * B * H;
        // the carried state is updated IN PLACE: a cell's c is read and written by the same thread, and the recurrent
        // product reads the bf16 copy of h (HB / the previous step's y row), never h_state itself
        for (int t = 0; t < Tl; ++t) {
            const dim3 grid1((B + 15) / 16, H / 16), grid4((B + 63) / 64, H / 16);
            const bf16_t* hb = t == 0 ? HB + (size_t)l * B * H : Y + (size_t)(t - 1) * H;
            const long long ldh = t == 0 ? (long long)H : (long long)Tl * H;
            if (B <= 16)
                hipLaunchKernelGGL(enc_lstm_step<1>, grid1, dim3(512), 0, s, X + (size_t)t * K, (long long)Tl * K, K,
                                   (const bf16_t*)w_ih[l], (const bf16_t*)w_hh[l], b_ih[l], b_hh[l], hb, ldh, cl, hl, cl,
                                   Y + (size_t)t * H, (long long)Tl * H, B, H);
            else
                hipLaunchKernelGGL(enc_lstm_tile, grid4, dim3(256), 0, s, X + (size_t)t * K, (long long)Tl * K, K,
                                   (const bf16_t*)w_ih[l], (const bf16_t*)w_hh[l], b_ih[l], b_hh[l], hb, ldh, cl, hl, cl,
                                   Y + (size_t)t * H, (long long)Tl * H, B, H);
        }
        const int Tn = (Tl + reduce[l] - 1) / reduce[l];
        void* dst = (l == L - 1) ? out : (void*)Xn;
        if ((rc = edgedict_layernorm_fwd(ED_BF16, Y, l > 0 ? X : nullptr, ln_gamma[l], ln_beta[l], dst, mean, rstd, B, Tl,
                                         H, reduce[l], 1e-5f, s)))
            return rc;
        bf16_t* tmp = X; X = Xn; Xn = tmp;
        Tl = Tn;
        K = H;
    }
    *T_out = Tl;
    ED_CHECK_LAUNCH("stream_encoder_step");
    return ED_OK;
}

// shapes the fused frame covers (otherwise decode.hip composes the frame from the general kernels)
bool ed_decode_fused_ok(int dtype, int emb_dtype, int J, int V, int E, int H, int P2) {
    static const int on = [] { const char* e = getenv("EDGEDICT_DECODE_FUSED"); return e ? atoi(e) : 1; }();
    return on && (dtype == ED_BF16 || (dtype == ED_F32 && emb_dtype == ED_F32)) && J % 32 == 0 && P2 % 32 == 0 &&
           E % 32 == 0 && H % 32 == 0 && V % 4 == 0;
}
// slice partials of one frame: [B][ceil(V / 64)] PickPart (32 bytes each: V / 2 bytes per row - they fit the fp32
// logits buffer of the composed path, which the fused frame never fills)
size_t ed_decode_fused_ws_bytes(int B, int V) {
    return align256((size_t)B * ((V + 63) / 64) * sizeof(PickPart));
}

namespace {

template <typename ET>
void launch_lstm_steps(int first_mode, const PickPart* parts, int NS, int unk, int blank, int32_t* pred, int32_t* tokens,
                       long long tok_stride, int t, float* score, const void* emb, int emb_dtype, int E, int L,
                       const void* const* w_ih, const void* const* w_hh, const float* const* b_ih,
                       const float* const* b_hh, int H, const float* h_state, const float* c_state, float* h_new,
                       float* c_new, void* const* Y, int B, int V, hipStream_t s) {
    const int RB = (B + 15) / 16;
    for (int k = 0; k < L; ++k) {
        LstmStepArgs<ET> A;
        A.V = V;
        A.parts = parts; A.nslices = NS; A.unk = unk; A.blank = blank;
        A.pred = pred; A.tokens = tokens; A.tok_stride = tok_stride; A.t = t; A.score = score;
        A.emb = emb; A.emb_f32 = emb_dtype == ED_F32 ? 1 : 0; A.E = E;
        A.x_prev = k > 0 ? (const ET*)Y[(k - 1) & 1] : nullptr;
        A.w_ih = (const ET*)w_ih[k]; A.w_hh = (const ET*)w_hh[k]; A.b_ih = b_ih[k]; A.b_hh = b_hh[k];
        A.h_in = h_state + (size_t)k * B * H; A.c_in = c_state + (size_t)k * B * H;
        A.h_out = h_new + (size_t)k * B * H; A.c_out = c_new + (size_t)k * B * H;
        A.y_out = (ET*)Y[k & 1];
        A.B = B; A.H = H; A.Kx = k == 0 ? E : H;
        hipLaunchKernelGGL(dec_lstm_step<ET>, dim3(RB, H / 16), dim3(256), 0, s, A, k == 0 ? first_mode : 0);
    }
}

template <typename ET>
int beam_step_t(const void* E1t, long long e_row_stride, int B, int J, const void* W1d, long long ldw1, const float* b1,
                int P2, const void* W2, const float* b2, int V, const void* emb, int emb_dtype, int E, int L,
                const void* const* w_ih, const void* const* w_hh, const float* const* b_ih, const float* const* b_hh,
                int H, const void* Wp, const float* bp, const float* h_state, const float* c_state, const int32_t* pred,
                void* dec_new, void* hid, float* logits, float* h_new, float* c_new, void* Y0, void* Y1, hipStream_t s) {
    const int RB = (B + 15) / 16;
    void* Y[2] = {Y0, Y1};
    launch_lstm_steps<ET>(2, nullptr, 0, -1, -1, const_cast<int32_t*>(pred), nullptr, 0, 0, nullptr, emb, emb_dtype, E, L,
                          w_ih, w_hh, b_ih, b_hh, H, h_state, c_state, h_new, c_new, Y, B, V, s);
    hipLaunchKernelGGL(dec_proj_commit<ET>, dim3(RB, (P2 + 63) / 64), dim3(256), 0, s, (const ET*)Y[(L - 1) & 1], H,
                       (const ET*)Wp, bp, P2, (const int32_t*)nullptr, 0, (ET*)dec_new, (float*)nullptr,
                       (const float*)nullptr, (float*)nullptr, (const float*)nullptr, L, B);
    hipLaunchKernelGGL(dec_joint_hidden<ET>, dim3(RB, (J + 63) / 64), dim3(256), 0, s, (const ET*)E1t, e_row_stride,
                       (const ET*)dec_new, P2, (const ET*)W1d, ldw1, b1, (ET*)hid, B, J);
    hipLaunchKernelGGL(dec_logits_pick<ET>, dim3(RB, (V + 63) / 64), dim3(256), 0, s, (const ET*)hid, J, (const ET*)W2,
                       b2, V, -1, 0, (PickPart*)nullptr, logits, B);
    return ED_OK;
}

template <typename ET>
int frame_t(const void* E1t, long long e_row_stride, int B, int J, const void* W1d, long long ldw1, const float* b1,
            int P2, const void* W2, const float* b2, int V, const void* emb, int emb_dtype, int E, int L,
            const void* const* w_ih, const void* const* w_hh, const float* const* b_ih, const float* const* b_hh, int H,
            const void* Wp, const float* bp, float* h_state, float* c_state, void* dec_out, int blank, int unk,
            int32_t* tokens_out, long long tok_stride, int t, float* score, void* hid, void* parts, int32_t* pred,
            float* h_new, float* c_new, void* Y0, void* Y1, hipStream_t s) {
    const int RB = (B + 15) / 16, NS = (V + 63) / 64;
    hipLaunchKernelGGL(dec_joint_hidden<ET>, dim3(RB, (J + 63) / 64), dim3(256), 0, s, (const ET*)E1t, e_row_stride,
                       (const ET*)dec_out, P2, (const ET*)W1d, ldw1, b1, (ET*)hid, B, J);
    hipLaunchKernelGGL(dec_logits_pick<ET>, dim3(RB, NS), dim3(256), 0, s, (const ET*)hid, J, (const ET*)W2, b2, V,
                       unk, score ? 1 : 0, (PickPart*)parts, (float*)nullptr, B);
    void* Y[2] = {Y0, Y1};
    launch_lstm_steps<ET>(1, (const PickPart*)parts, NS, unk, blank, pred, tokens_out, tok_stride, t, score, emb, emb_dtype,
                          E, L, w_ih, w_hh, b_ih, b_hh, H, h_state, c_state, h_new, c_new, Y, B, V, s);
    hipLaunchKernelGGL(dec_proj_commit<ET>, dim3(RB, (P2 + 63) / 64), dim3(256), 0, s, (const ET*)Y[(L - 1) & 1], H,
                       (const ET*)Wp, bp, P2, pred, blank, (ET*)dec_out, h_state, h_new, c_state, c_new, L, B);
    return ED_OK;
}

}  // namespace

// beam search (decode.hip): prediction-network step on pred[b] from (h_state, c_state) -> h_new / c_new, dec_new, the
// joint's hidden vector of frame t and the logits (fp32 [B, V]) of every row - 3 + L launches instead of 5 + 2 L
int ed_decode_fused_beam_step(int dtype, const void* E1t, long long e_row_stride, int B, int J, const void* W1d,
                              long long ldw1, const float* b1, int P2, const void* W2, const float* b2, int V,
                              const void* emb, int emb_dtype, int E, int L, const void* const* w_ih,
                              const void* const* w_hh, const float* const* b_ih, const float* const* b_hh, int H,
                              const void* Wp, const float* bp, const float* h_state, const float* c_state,
                              const int32_t* pred, void* dec_new, void* hid, float* logits, float* h_new, float* c_new,
                              void* Y0, void* Y1, hipStream_t s) {
    if (dtype == ED_F32)
        beam_step_t<float>(E1t, e_row_stride, B, J, W1d, ldw1, b1, P2, W2, b2, V, emb, emb_dtype, E, L, w_ih, w_hh, b_ih,
                           b_hh, H, Wp, bp, h_state, c_state, pred, dec_new, hid, logits, h_new, c_new, Y0, Y1, s);
    else
        beam_step_t<bf16_t>(E1t, e_row_stride, B, J, W1d, ldw1, b1, P2, W2, b2, V, emb, emb_dtype, E, L, w_ih, w_hh, b_ih,
                            b_hh, H, Wp, bp, h_state, c_state, pred, dec_new, hid, logits, h_new, c_new, Y0, Y1, s);
    ED_CHECK_LAUNCH("decode_fused_beam_step");
    return ED_OK;
}

// one frame of the search: see the file header.  hid [B, J], parts = ed_decode_fused_ws_bytes(B, V) bytes,
// h_new / c_new [L, B, H] fp32, Y [2][B, H] (ping-pong between layers); pred [B] out
int ed_decode_fused_frame(int dtype, const void* E1t, long long e_row_stride, int B, int J, const void* W1d,
                          long long ldw1, const float* b1, int P2, const void* W2, const float* b2, int V,
                          const void* emb, int emb_dtype, int E, int L, const void* const* w_ih,
                          const void* const* w_hh, const float* const* b_ih, const float* const* b_hh, int H,
                          const void* Wp, const float* bp, float* h_state, float* c_state, void* dec_out, int blank,
                          int unk, int32_t* tokens_out, long long tok_stride, int t, float* score, void* hid, void* parts,
                          int32_t* pred, float* h_new, float* c_new, void* Y0, void* Y1, hipStream_t s) {
    if (dtype == ED_F32)
        frame_t<float>(E1t, e_row_stride, B, J, W1d, ldw1, b1, P2, W2, b2, V, emb, emb_dtype, E, L, w_ih, w_hh, b_ih, b_hh,
                       H, Wp, bp, h_state, c_state, dec_out, blank, unk, tokens_out, tok_stride, t, score, hid, parts,
                       pred, h_new, c_new, Y0, Y1, s);
    else
        frame_t<bf16_t>(E1t, e_row_stride, B, J, W1d, ldw1, b1, P2, W2, b2, V, emb, emb_dtype, E, L, w_ih, w_hh, b_ih, b_hh,
                        H, Wp, bp, h_state, c_state, dec_out, blank, unk, tokens_out, tok_stride, t, score, hid, parts,
                        pred, h_new, c_new, Y0, Y1, s);
    ED_CHECK_LAUNCH("decode_fused_frame");
    return ED_OK;
}
