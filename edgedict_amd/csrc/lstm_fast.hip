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
//   W_hh  (fwd)  frag[ub][gate][ks][lane][8]  n = lane & 15 -> W_hh row gate*H + ub*16 + n, k as above
//   W_hh^T(bwd)  frag[ub][ks][lane][8]   n = lane & 15 -> hidden unit ub*16 + n, k = gate-row index
//
// The step kernel that PRODUCES h_t (or dG_t) writes it twice: in the plain [B,T,*] layout the
// batched GEMMs need, and into the ping-pong fragment buffer the NEXT step's MFMA reads.
// The weight images are rebuilt from the fp32 master weights once per optimiser step
// (edgedict_lstm_pack_weights).  All loads of a wave's K slice are issued before the first MFMA
// (up to 40 x 16 B in flight per lane), so L2 latency is paid once per step.
#include "common.hpp"
#include "lstm_fast.hpp"

#ifndef ED_LSTM_DBG
#define ED_LSTM_DBG 0   // tools/lstm_step_probe.hip builds ablation variants with this mask
#endif

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
// fwd image: [H/16][4 gates][H/32][64][8];  bwd image: [H/16][4H/32][64][8]
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
            const int ks = (int)(blk % KSf);
            const int gate = (int)((blk / KSf) & 3);
            const int ub = (int)(blk / (4 * KSf));
            const int row = gate * H + ub * 16 + (lane & 15);
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
// workgroup = (16 hidden units x 4 gates = four 16-wide N tiles, 16 batch rows = one M tile);
// K = H split over the 4 waves.  Results are staged in LDS and leave the CU as 16-byte stores.
constexpr int CH = 8;  // k-steps loaded per batch (8 x (1 A + 4 W) x 16 B = 640 B in flight / lane)

struct __attribute__((aligned(16))) FwdStage {
    bf16_t g[4][16][16];   // post-activation gates
    bf16_t h[16][16];
    float c[16][16];
};

__global__ __launch_bounds__(256) void lstm_step_fwd_fast(
    bf16_t* __restrict__ G, const bf16_t* __restrict__ hfrag_in, bf16_t* __restrict__ hfrag_out,
    bf16_t* __restrict__ Hprev, bf16_t* __restrict__ Y, float* __restrict__ Cst,
    const bf16_t* __restrict__ Wfrag, const float* __restrict__ c0, float* __restrict__ hN,
    float* __restrict__ cN, int B, int Tn, int H, int t, int mt_base) {
    __shared__ __attribute__((aligned(16))) float red[4][4][16][17];
    __shared__ FwdStage st;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ub = blockIdx.x, mt = blockIdx.y + mt_base;
    const int KS = H >> 5;
    const int per = (KS + 3) >> 2;
    const int ks_beg = wave * per, ks_end = min(KS, ks_beg + per);

    // pointwise operands of this thread (row r, unit u), fetched early: raw bits, converted later
    const int r = threadIdx.x >> 4, u = threadIdx.x & 15;
    const int b = mt * 16 + r, j = ub * 16 + u;
    const bool live = b < B;
    const long long row = (long long)(live ? b : 0) * Tn + t;
    bf16_t* grow = G + row * 4 * H;
    bf16_t graw[4] = {0, 0, 0, 0};
    float cprev = 0.f;
    if (live && !(ED_LSTM_DBG & 16)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) graw[g] = grow[g * H + j];
        cprev = (t > 0) ? Cst[(row - 1) * H + j] : (c0 ? c0[(long long)b * H + j] : 0.f);
    }

    f32x4_t acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bf16_t* wbase = Wfrag + ((long long)ub * 4 * KS * 64 + lane) * 8;
    const bf16_t* abase = hfrag_in + ((long long)mt * KS * 64 + lane) * 8;
    for (int ks0 = ks_beg; ks0 < ks_end && !(ED_LSTM_DBG & 1); ks0 += CH) {
        bf16x8_t a[CH], w[CH][4];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int ks = min(ks0 + i, ks_end - 1);   // clamp: duplicates are masked below
            a[i] = ldfrag(abase + (long long)ks * 512);
#pragma unroll
            for (int g = 0; g < 4; ++g) w[i][g] = ldfrag(wbase + ((long long)g * KS + ks) * 512);
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (ks0 + i < ks_end) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], w[i][g], acc[g], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[wave][g][(lane >> 4) * 4 + q][lane & 15] = acc[g][q];
    __syncthreads();
    float pre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
        pre[g] = bf16_to_f32(graw[g]) + red[0][g][r][u] + red[1][g][r][u] + red[2][g][r][u] +
                 red[3][g][r][u];
    float ig, fg, gg, og, c, h;
    if (ED_LSTM_DBG & 8) {
        ig = pre[0] * 0.5f; fg = pre[1] * 0.25f; gg = pre[2] + 1.f; og = pre[3] - 1.f;
        c = fg * cprev + ig * gg;
        h = og * c;
    } else {
        ig = sigmoidf_(pre[0]); fg = sigmoidf_(pre[1]); gg = tanhf(pre[2]); og = sigmoidf_(pre[3]);
        c = fg * cprev + ig * gg;
        h = og * tanhf(c);
    }
    st.g[0][r][u] = f32_to_bf16(ig);
    st.g[1][r][u] = f32_to_bf16(fg);
    st.g[2][r][u] = f32_to_bf16(gg);
    st.g[3][r][u] = f32_to_bf16(og);
    st.h[r][u] = f32_to_bf16(h);
    st.c[r][u] = c;
    __syncthreads();
    // cooperative 16-byte stores: 128 (gates) + 32 (Y) + 32 (Hprev|final h) + 64 (c) + 32 (fragment)
    const bool last = (t + 1 == Tn);
    for (int task = threadIdx.x; task < 288; task += 256) {
        if (task < 128) {
            if (ED_LSTM_DBG & 2) continue;
            const int g = task >> 5, rr = (task >> 1) & 15, hf = task & 1;
            if (mt * 16 + rr >= B) continue;
            *reinterpret_cast<uint4*>(G + (((long long)(mt * 16 + rr) * Tn + t) * 4 + g) * H +
                                      ub * 16 + hf * 8) =
                *reinterpret_cast<const uint4*>(&st.g[g][rr][hf * 8]);
        } else if (ED_LSTM_DBG & 4) {
            continue;
        } else if (task < 192) {
            const int which = (task - 128) >> 5, rr = ((task - 128) >> 1) & 15, hf = task & 1;
            const int bb = mt * 16 + rr;
            if (bb >= B) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(&st.h[rr][hf * 8]);
            const long long ro = (long long)bb * Tn + t;
            if (which == 0) {
                *reinterpret_cast<uint4*>(Y + ro * H + ub * 16 + hf * 8) = v;
            } else if (!last) {
                *reinterpret_cast<uint4*>(Hprev + (ro + 1) * H + ub * 16 + hf * 8) = v;
            } else if (hN) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    hN[(long long)bb * H + ub * 16 + hf * 8 + e] = bf16_to_f32(st.h[rr][hf * 8 + e]);
            }
        } else if (task < 256) {
            const int rr = (task - 192) >> 2, q = task & 3;
            const int bb = mt * 16 + rr;
            if (bb >= B) continue;
            const float4 v = *reinterpret_cast<const float4*>(&st.c[rr][q * 4]);
            *reinterpret_cast<float4*>(Cst + ((long long)bb * Tn + t) * H + ub * 16 + q * 4) = v;
            if (last && cN) *reinterpret_cast<float4*>(cN + (long long)bb * H + ub * 16 + q * 4) = v;
        } else if (!last) {
            // fragment image for the next step: k = ub*16 + hf*8 + e, 16 rows x 16 B contiguous
            const int hf = (task - 256) >> 4, rr = task & 15;
            const int k = ub * 16 + hf * 8;
            *reinterpret_cast<uint4*>(hfrag_out + frag_off(mt * 16 + rr, k, KS)) =
                *reinterpret_cast<const uint4*>(&st.h[rr][hf * 8]);
        }
    }
}

// ------------------------------------------------------------------ backward step
// workgroup = (16 hidden units, 16 batch rows); K = 4H, one quarter per wave, two accumulators
struct __attribute__((aligned(16))) BwdStage {
    bf16_t d[4][16][16];   // dL/d(pre-activation) per gate
};

__global__ __launch_bounds__(256) void lstm_step_bwd_fast(
    bf16_t* __restrict__ G, const bf16_t* __restrict__ gfrag_in, bf16_t* __restrict__ gfrag_out,
    const bf16_t* __restrict__ dY, const float* __restrict__ Cst, const float* __restrict__ c0,
    const bf16_t* __restrict__ WTfrag, float* __restrict__ dC, int B, int Tn, int H, int t,
    int mt_base) {
    __shared__ float red[4][16][17];
    __shared__ BwdStage st;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ub = blockIdx.x, mt = blockIdx.y + mt_base;
    const int KS = H >> 3;  // 4H / 32
    const int per = (KS + 3) >> 2;
    const int ks_beg = wave * per, ks_end = min(KS, ks_beg + per);

    const int bl = threadIdx.x >> 4, nl = threadIdx.x & 15;
    const int b = mt * 16 + bl, j = ub * 16 + nl;
    const bool live = b < B && j < H;
    const long long row = (long long)(live ? b : 0) * Tn + t;
    bf16_t* grow = G + row * 4 * H;
    bf16_t graw[4] = {0, 0, 0, 0}, dyraw = 0;
    float c = 0.f, cprev = 0.f, dcn = 0.f;
    if (live) {
#pragma unroll
        for (int g = 0; g < 4; ++g) graw[g] = grow[g * H + j];
        c = Cst[row * H + j];
        cprev = (t > 0) ? Cst[(row - 1) * H + j] : (c0 ? c0[(long long)b * H + j] : 0.f);
        dcn = dC[(long long)b * H + j];
        if (dY) dyraw = dY[row * H + j];
    }

    f32x4_t acc0 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    if (t + 1 < Tn) {
        const bf16_t* abase = gfrag_in + ((long long)mt * KS * 64 + lane) * 8;
        const bf16_t* wbase = WTfrag + ((long long)ub * KS * 64 + lane) * 8;
        for (int ks0 = ks_beg; ks0 < ks_end; ks0 += 2 * CH) {
            bf16x8_t a[2 * CH], w[2 * CH];
#pragma unroll
            for (int i = 0; i < 2 * CH; ++i) {
                const int ks = min(ks0 + i, ks_end - 1);
                a[i] = ldfrag(abase + (long long)ks * 512);
                w[i] = ldfrag(wbase + (long long)ks * 512);
            }
#pragma unroll
            for (int i = 0; i < 2 * CH; i += 2) {
                if (ks0 + i < ks_end)
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], w[i], acc0, 0, 0, 0);
                if (ks0 + i + 1 < ks_end)
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i + 1], w[i + 1], acc1, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) red[wave][(lane >> 4) * 4 + q][lane & 15] = acc0[q] + acc1[q];
    __syncthreads();
    const float ig = bf16_to_f32(graw[0]), fg = bf16_to_f32(graw[1]), gg = bf16_to_f32(graw[2]),
                og = bf16_to_f32(graw[3]);
    const float dh = bf16_to_f32(dyraw) + red[0][bl][nl] + red[1][bl][nl] + red[2][bl][nl] +
                     red[3][bl][nl];
    const float tc = tanhf(c);
    const float dct = dcn + dh * og * (1.f - tc * tc);
    st.d[0][bl][nl] = f32_to_bf16(dct * gg * ig * (1.f - ig));
    st.d[1][bl][nl] = f32_to_bf16(dct * cprev * fg * (1.f - fg));
    st.d[2][bl][nl] = f32_to_bf16(dct * ig * (1.f - gg * gg));
    st.d[3][bl][nl] = f32_to_bf16(dh * tc * og * (1.f - og));
    if (live) dC[(long long)b * H + j] = dct * fg;
    __syncthreads();
    // 128 x 16-byte stores into G (plain layout) + 128 into the fragment image (4 gates x 2 x 16 rows)
    {
        const int task = threadIdx.x & 127;
        const int g = task >> 5, rr = (task >> 1) & 15, hf = task & 1;
        const int bb = mt * 16 + rr;
        const uint4 v = *reinterpret_cast<const uint4*>(&st.d[g][rr][hf * 8]);
        if (threadIdx.x < 128) {
            if (bb < B)
                *reinterpret_cast<uint4*>(G + (((long long)bb * Tn + t) * 4 + g) * H + ub * 16 + hf * 8) = v;
        } else if (t > 0) {
            *reinterpret_cast<uint4*>(gfrag_out + frag_off(bb, g * H + ub * 16 + hf * 8, KS)) = v;
        }
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
    dim3 grid(H / 16, (B + 15) / 16);
    for (int t = 0; t < Tn; ++t)
        hipLaunchKernelGGL(lstm_step_fwd_fast, grid, dim3(256), 0, s, (bf16_t*)G, frag[t & 1],
                           frag[(t + 1) & 1], (bf16_t*)Hprev, (bf16_t*)Y, Cst, (const bf16_t*)Wfrag,
                           c0, hN, cN, B, Tn, H, t, 0);
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
                           H, t, 0);
    ED_CHECK_LAUNCH("lstm_step_bwd_fast");
    return ED_OK;
}

// One independent launch chain per 16-row tile (the recurrences of different batch rows never
// interact), each on its own stream: used by the graph path and by tools/lstm_step_probe.hip.
int ed_lstm_fwd_fast_chains(void* G, void* Hprev, void* Y, float* Cst, const void* Wfrag,
                            const float* c0, float* hN, float* cN, int B, int Tn, int H, void* ws,
                            hipStream_t* streams, int nstreams) {
    const size_t half = (frag_bytes(B, 4 * H) + 255) / 256 * 256;
    bf16_t* frag[2] = {(bf16_t*)ws, (bf16_t*)((char*)ws + half)};
    const int MT = (B + 15) / 16;
    for (int t = 0; t < Tn; ++t)
        for (int m = 0; m < MT; ++m)
            hipLaunchKernelGGL(lstm_step_fwd_fast, dim3(H / 16, 1), dim3(256), 0,
                               streams[m % nstreams], (bf16_t*)G, frag[t & 1], frag[(t + 1) & 1],
                               (bf16_t*)Hprev, (bf16_t*)Y, Cst, (const bf16_t*)Wfrag, c0, hN, cN, B,
                               Tn, H, t, m);
    ED_CHECK_LAUNCH("lstm_step_fwd_fast chains");
    return ED_OK;
}
