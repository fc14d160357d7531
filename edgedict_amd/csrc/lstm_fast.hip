// bf16 fast path of the LSTM recurrence (see lstm.hip for the arithmetic and the generic path).
//
// Why a second path: in the generic step kernel every MFMA operand is fetched from L2 in the
// fragment shape (lane&15 = row, 16 B per lane), i.e. each wave-wide load touches 16 different
// rows x 64 B.  On CDNA4 the texture-addresser then needs ~64 tag look-ups per instruction and the
// loads of one k-step are consumed before the next are issued: 14 us per step at H = 1024, almost
// all of it waiting on L2.  Here every recurrent operand lives in HBM/L2 already in MFMA
// FRAGMENT ORDER, so a wave-wide 16-byte load is one contiguous 1 KiB burst that lands directly in
// the registers the MFMA reads -- no LDS staging, no shuffles:
//
//   activations  frag[mt][ks][lane][8]   row = mt*16 + (lane & 15),  k = ks*32 + (lane >> 4)*8 + e
//   W_hh  (fwd)  frag[jb][ks][lane][8]   n = lane & 15 -> W_hh row (n/4)*H + jb*4 + (n%4), k as above
//   W_hh^T(bwd)  frag[ub][ks][lane][8]   n = lane & 15 -> hidden unit ub*16 + n, k = gate-row index
//
// The step kernel that PRODUCES h_t (or dG_t) writes it twice: in the plain [B,T,*] layout the
// batched GEMMs need, and into the ping-pong fragment buffer the NEXT step's MFMA reads.
// The weight images are rebuilt from the fp32 master weights once per optimiser step
// (edgedict_lstm_pack_weights).  All loads of a wave's K slice are issued before the first MFMA
// (up to 40 x 16 B in flight per lane), so L2 latency is paid once per step.
#include "common.hpp"
#include "lstm_fast.hpp"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ bf16x8_t ldfrag(const bf16_t* p) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    return *reinterpret_cast<bf16x8_t*>(&v);
}
__device__ __forceinline__ bf16x8_t zfrag() {
    uint4 v = make_uint4(0, 0, 0, 0);
    return *reinterpret_cast<bf16x8_t*>(&v);
}

// element (row, k) of an activation fragment image with KS k-steps per row tile
__device__ __forceinline__ long long frag_off(int row, int k, int KS) {
    return ((((long long)(row >> 4) * KS + (k >> 5)) * 64) + ((k & 31) >> 3) * 16 + (row & 15)) * 8 +
           (k & 7);
}

// ------------------------------------------------------------------ weight packing
// fwd image: [H/4][H/32][64][8];  bwd image: [H/16][4H/32][64][8]
template <typename TS>
__global__ void pack_whh(const TS* __restrict__ W, bf16_t* __restrict__ fwd,
                         bf16_t* __restrict__ bwd, int H) {
    const long long n = (long long)4 * H * H;
    const int KSf = H / 32, KSb = H / 8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7);
        const int lane = (int)((i >> 3) & 63);
        const long long blk = i >> 9;
        if (fwd) {
            const int ks = (int)(blk % KSf), jb = (int)(blk / KSf);
            const int nn = lane & 15;
            const int row = (nn >> 2) * H + jb * 4 + (nn & 3);
            const int k = ks * 32 + (lane >> 4) * 8 + e;
            fwd[i] = f32_to_bf16(ElemIO<TS>::load(W + (long long)row * H + k));
        }
        if (bwd) {
            const int ks = (int)(blk % KSb), ub = (int)(blk / KSb);
            const int j = ub * 16 + (lane & 15);
            const int k = ks * 32 + (lane >> 4) * 8 + e;  // gate-row index in [0, 4H)
            bwd[i] = f32_to_bf16(ElemIO<TS>::load(W + (long long)k * H + j));
        }
    }
}

// h0 -> Hprev[:, 0, :] (plain) and the first fragment image
__global__ void init_h_fast(bf16_t* __restrict__ Hprev, bf16_t* __restrict__ hfrag,
                            const float* __restrict__ h0, int B, int Tn, int H) {
    const long long n = (long long)B * H;
    const int KS = H / 32;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / H), j = (int)(i % H);
        const bf16_t v = f32_to_bf16(h0 ? h0[i] : 0.f);
        Hprev[((long long)b * Tn) * H + j] = v;
        hfrag[frag_off(b, j, KS)] = v;
    }
}

// ------------------------------------------------------------------ forward step
constexpr int CH = 8;  // k-steps loaded per batch (8 x (4 A + 1 W) x 16 B = 640 B in flight / lane)

__global__ __launch_bounds__(256) void lstm_step_fwd_fast(
    bf16_t* __restrict__ G, const bf16_t* __restrict__ hfrag_in, bf16_t* __restrict__ hfrag_out,
    bf16_t* __restrict__ Hprev, bf16_t* __restrict__ Y, float* __restrict__ Cst,
    const bf16_t* __restrict__ Wfrag, const float* __restrict__ c0, float* __restrict__ hN,
    float* __restrict__ cN, int B, int Tn, int H, int t) {
    __shared__ float red[4][64][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int jb = blockIdx.x;           // 4 hidden units
    const int mt0 = blockIdx.y * 4;      // 4 row tiles of 16
    const int MT = (B + 15) >> 4;
    const int KS = H >> 5;
    const int per = (KS + 3) >> 2;
    const int ks_beg = wave * per, ks_end = min(KS, ks_beg + per);

    // pointwise operands of this thread, fetched early so they overlap the MFMA phase
    const int bl = threadIdx.x >> 2, u = threadIdx.x & 3;
    const int b = blockIdx.y * 64 + bl, j = jb * 4 + u;
    const bool live = b < B;
    float pre[4] = {0.f, 0.f, 0.f, 0.f};
    float cprev = 0.f;
    bf16_t* grow = G + ((long long)(live ? b : 0) * Tn + t) * 4 * H;
    if (live) {
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[g] = bf16_to_f32(grow[g * H + j]);
        cprev = (t > 0) ? Cst[((long long)b * Tn + t - 1) * H + j]
                        : (c0 ? c0[(long long)b * H + j] : 0.f);
    }

    f32x4_t acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bf16_t* wbase = Wfrag + ((long long)jb * KS * 64 + lane) * 8;
    const bf16_t* abase = hfrag_in + (long long)lane * 8;
    for (int ks0 = ks_beg; ks0 < ks_end; ks0 += CH) {
        bf16x8_t w[CH], a[CH][4];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int ks = ks0 + i;
            const bool ok = ks < ks_end;
            w[i] = ok ? ldfrag(wbase + (long long)ks * 512) : zfrag();
#pragma unroll
            for (int m = 0; m < 4; ++m)
                a[i][m] = (ok && mt0 + m < MT)
                              ? ldfrag(abase + ((long long)(mt0 + m) * KS + ks) * 512)
                              : zfrag();
        }
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int m = 0; m < 4; ++m)
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][m], w[i], acc[m], 0, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            red[wave][m * 16 + (lane >> 4) * 4 + r][lane & 15] = acc[m][r];
    __syncthreads();
    if (!live) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int col = g * 4 + u;
        pre[g] += red[0][bl][col] + red[1][bl][col] + red[2][bl][col] + red[3][bl][col];
    }
    const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]),
                og = sigmoidf_(pre[3]);
    const float c = fg * cprev + ig * gg;
    const float h = og * tanhf(c);
    grow[j] = f32_to_bf16(ig);
    grow[H + j] = f32_to_bf16(fg);
    grow[2 * H + j] = f32_to_bf16(gg);
    grow[3 * H + j] = f32_to_bf16(og);
    const long long row = (long long)b * Tn + t;
    Cst[row * H + j] = c;
    const bf16_t hb = f32_to_bf16(h);
    Y[row * H + j] = hb;
    if (t + 1 < Tn) {
        Hprev[(row + 1) * H + j] = hb;
        hfrag_out[frag_off(b, j, KS)] = hb;
    } else {
        if (hN) hN[(long long)b * H + j] = h;
        if (cN) cN[(long long)b * H + j] = c;
    }
}

// ------------------------------------------------------------------ backward step
// workgroup = (16 hidden units, 16 batch rows); K = 4H, one quarter per wave, two accumulators
__global__ __launch_bounds__(256) void lstm_step_bwd_fast(
    bf16_t* __restrict__ G, const bf16_t* __restrict__ gfrag_in, bf16_t* __restrict__ gfrag_out,
    const bf16_t* __restrict__ dY, const float* __restrict__ Cst, const float* __restrict__ c0,
    const bf16_t* __restrict__ WTfrag, float* __restrict__ dC, int B, int Tn, int H, int t) {
    __shared__ float red[4][16][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ub = blockIdx.x, mt = blockIdx.y;
    const int KS = H >> 3;  // 4H / 32
    const int per = (KS + 3) >> 2;
    const int ks_beg = wave * per, ks_end = min(KS, ks_beg + per);

    const int bl = threadIdx.x >> 4, nl = threadIdx.x & 15;
    const int b = mt * 16 + bl, j = ub * 16 + nl;
    const bool live = b < B && j < H;
    const long long row = (long long)(live ? b : 0) * Tn + t;
    bf16_t* grow = G + row * 4 * H;
    float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, c = 0.f, cprev = 0.f, dcn = 0.f, dy = 0.f;
    if (live) {
        ig = bf16_to_f32(grow[j]);
        fg = bf16_to_f32(grow[H + j]);
        gg = bf16_to_f32(grow[2 * H + j]);
        og = bf16_to_f32(grow[3 * H + j]);
        c = Cst[row * H + j];
        cprev = (t > 0) ? Cst[(row - 1) * H + j] : (c0 ? c0[(long long)b * H + j] : 0.f);
        dcn = dC[(long long)b * H + j];
        if (dY) dy = bf16_to_f32(dY[row * H + j]);
    }

    f32x4_t acc0 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    if (t + 1 < Tn) {
        const bf16_t* abase = gfrag_in + ((long long)mt * KS * 64 + lane) * 8;
        const bf16_t* wbase = WTfrag + ((long long)ub * KS * 64 + lane) * 8;
        for (int ks0 = ks_beg; ks0 < ks_end; ks0 += 2 * CH) {
            bf16x8_t a[2 * CH], w[2 * CH];
#pragma unroll
            for (int i = 0; i < 2 * CH; ++i) {
                const int ks = ks0 + i;
                const bool ok = ks < ks_end;
                a[i] = ok ? ldfrag(abase + (long long)ks * 512) : zfrag();
                w[i] = ok ? ldfrag(wbase + (long long)ks * 512) : zfrag();
            }
#pragma unroll
            for (int i = 0; i < 2 * CH; i += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], w[i], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i + 1], w[i + 1], acc1, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(lane >> 4) * 4 + r][lane & 15] = acc0[r] + acc1[r];
    __syncthreads();
    if (!live) return;
    const float dh = dy + red[0][bl][nl] + red[1][bl][nl] + red[2][bl][nl] + red[3][bl][nl];
    const float tc = tanhf(c);
    const float dct = dcn + dh * og * (1.f - tc * tc);
    const bf16_t di = f32_to_bf16(dct * gg * ig * (1.f - ig));
    const bf16_t df = f32_to_bf16(dct * cprev * fg * (1.f - fg));
    const bf16_t dg = f32_to_bf16(dct * ig * (1.f - gg * gg));
    const bf16_t dob = f32_to_bf16(dh * tc * og * (1.f - og));
    grow[j] = di;
    grow[H + j] = df;
    grow[2 * H + j] = dg;
    grow[3 * H + j] = dob;
    dC[(long long)b * H + j] = dct * fg;
    if (t > 0) {
        gfrag_out[frag_off(b, j, KS)] = di;
        gfrag_out[frag_off(b, H + j, KS)] = df;
        gfrag_out[frag_off(b, 2 * H + j, KS)] = dg;
        gfrag_out[frag_off(b, 3 * H + j, KS)] = dob;
    }
}

__global__ void fill_f32_fast(float* p, long long n, float v) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        p[i] = v;
}

inline size_t frag_bytes(int B, int K) { return (size_t)((B + 15) / 16 * 16) * K * sizeof(bf16_t); }

}  // namespace

bool ed_lstm_fast_ok(int dtype, int H) { return dtype == ED_BF16 && H % 32 == 0; }

size_t ed_lstm_fast_ws_bytes(int B, int H) {
    // two ping-pong fragment images; the backward ones ([B16, 4H]) are the larger
    return 2 * ((frag_bytes(B, 4 * H) + 255) / 256 * 256);
}

int ed_lstm_pack(int src_dtype, const void* Whh, void* fwd, void* bwd, int H, hipStream_t s) {
    const long long n = (long long)4 * H * H;
    const int grid = ed_grid_for(n, 256, 4096);
    if (src_dtype == ED_F32)
        hipLaunchKernelGGL(pack_whh<float>, dim3(grid), dim3(256), 0, s, (const float*)Whh,
                           (bf16_t*)fwd, (bf16_t*)bwd, H);
    else
        hipLaunchKernelGGL(pack_whh<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)Whh,
                           (bf16_t*)fwd, (bf16_t*)bwd, H);
    ED_CHECK_LAUNCH("lstm_pack_weights");
    return ED_OK;
}

int ed_lstm_fwd_fast(void* G, void* Hprev, void* Y, float* Cst, const void* Wfrag, const float* h0,
                     const float* c0, float* hN, float* cN, int B, int Tn, int H, void* ws,
                     hipStream_t s) {
    const size_t half = (frag_bytes(B, 4 * H) + 255) / 256 * 256;
    bf16_t* frag[2] = {(bf16_t*)ws, (bf16_t*)((char*)ws + half)};
    hipLaunchKernelGGL(init_h_fast, dim3(ed_grid_for((long long)B * H, 256)), dim3(256), 0, s,
                       (bf16_t*)Hprev, frag[0], h0, B, Tn, H);
    ED_CHECK_LAUNCH("lstm init_h_fast");
    dim3 grid(H / 4, (B + 63) / 64);
    for (int t = 0; t < Tn; ++t)
        hipLaunchKernelGGL(lstm_step_fwd_fast, grid, dim3(256), 0, s, (bf16_t*)G, frag[t & 1],
                           frag[(t + 1) & 1], (bf16_t*)Hprev, (bf16_t*)Y, Cst, (const bf16_t*)Wfrag,
                           c0, hN, cN, B, Tn, H, t);
    ED_CHECK_LAUNCH("lstm_step_fwd_fast");
    return ED_OK;
}

int ed_lstm_bwd_fast(void* G, const void* dY, const float* Cst, const float* c0, const void* WTfrag,
                     float* dC, int B, int Tn, int H, void* ws, hipStream_t s) {
    const size_t half = (frag_bytes(B, 4 * H) + 255) / 256 * 256;
    bf16_t* frag[2] = {(bf16_t*)ws, (bf16_t*)((char*)ws + half)};
    hipLaunchKernelGGL(fill_f32_fast, dim3(ed_grid_for((long long)B * H, 256)), dim3(256), 0, s, dC,
                       (long long)B * H, 0.f);
    ED_CHECK_LAUNCH("lstm dC init");
    dim3 grid((H + 15) / 16, (B + 15) / 16);
    for (int t = Tn - 1; t >= 0; --t)
        hipLaunchKernelGGL(lstm_step_bwd_fast, grid, dim3(256), 0, s, (bf16_t*)G, frag[(t + 1) & 1],
                           frag[t & 1], (const bf16_t*)dY, Cst, c0, (const bf16_t*)WTfrag, dC, B, Tn,
                           H, t);
    ED_CHECK_LAUNCH("lstm_step_bwd_fast");
    return ED_OK;
}
