// Weights-stationary recurrence kernels of the encoder stack (bf16 throughput mode, H = 1024).
//
// Reference arithmetic: the nn.LSTM of ResLayerNormLSTM.forward (rnnt/models.py:55-75; PyTorch gate
// order i,f,g,o; c_t = f c_{t-1} + i g; h_t = o tanh(c_t)) - the same per-step arithmetic as
// stack_kernels.hip.  What differs is WHERE the operands live.
//
// stack_fwd_kernel makes one launch per time step, and nothing survives a kernel boundary on chip:
// W_hh (8 MB bf16 per layer) is re-streamed from MALL/HBM on every one of the 1606 layer-steps
// (12.6 us per 4-layer-step launch).  Here ONE launch carries a whole CHUNK of
// frames of every runnable layer:
//
//   * a layer occupies the 32 CUs of ONE XCD (workgroups find their XCD with HW_REG_XCC_ID and take
//     a ticket; XCDs without a layer in this launch exit at once and stay free for the chunk
//     products on the side stream);
//   * CU j keeps the 128 gate columns of hidden units [32 j, 32 j + 32) x K = 1024 of W_hh in its
//     REGISTER FILE for the whole launch: 4 waves x 256 registers (64 fragments of 16 columns x 32 k);
//   * per time step every CU gathers h_{t-1} (128 KB, MFMA A-fragment order - the same image the
//     step kernels exchange) from that XCD's L2 into LDS, runs 256 MFMA 16x16x32 per wave against
//     its registers, finishes its 64 x 32 cells in registers (c_t never leaves them inside a
//     launch), writes its 4 KB slice of the next image with write-through stores and bumps the
//     layer's arrival counter; the other CUs of the XCD poll that counter (tools/wsr_probe.hip
//     measured the step: 9.7 us with all 8 XCDs busy, vs 19 us per frame for the step kernels).
//
// Every spin is bounded and ends in a give-up code (err[0]); a launch that gave up leaves garbage
// behind and the caller reports ED_ERR_LAUNCH.
#include "stack_kernels.hpp"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int WH = 1024, WCUS = 32, WUPC = 32, WKS = 32;
constexpr int W_IMG = 64 * WH * 2;            // 128 KB: h image for 4 row blocks
constexpr int W_GT = 64 * 128 * 2;            // 16 KB: this CU's gate columns of one frame
constexpr int W_HS = 64 * WUPC * 2;           // 4 KB
constexpr int W_CS = 64 * WUPC * 4;           // 8 KB
constexpr int W_LDS = W_IMG + W_GT + W_HS + W_CS;
#ifndef WSR_LN_IN_RECURRENCE
#define WSR_LN_IN_RECURRENCE 0   // 1: the recurrence CUs normalise their layer's frames (two rows each, from the
#endif                           // h image in LDS); 0: the workers do, before the product (measured faster: the
                                 // residual row's HBM latency lands on the recurrence's critical path otherwise)

__device__ __forceinline__ float wsigm(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float wtanh(float x) {
    const float xc = fminf(fmaxf(x, -15.f), 15.f);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * xc));
}

// W_hh [4H, H] fp32 -> register image [cu 32][wave 4][tile 2][ks 32][lane 64][8] bf16:
// tile 0 columns n = lane & 15: n < 8 -> gate i of unit 32 cu + 8 wave + n, n >= 8 -> gate f of unit n - 8;
// tile 1: g | o.  k = ks * 32 + (lane >> 4) * 8 + e.
__global__ void wsr_pack_fwd_kernel(const float* __restrict__ W, bf16_t* __restrict__ out) {
    const long long n = 4ll * WH * WH;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), ks = (int)((i >> 9) & 31);
        const int tn = (int)((i >> 14) & 1), wv = (int)((i >> 15) & 3), cu = (int)(i >> 17);
        const int nn = lane & 15, gate = tn * 2 + (nn >> 3);
        const int unit = cu * WUPC + wv * 8 + (nn & 7);
        const int k = ks * 32 + (lane >> 4) * 8 + e;
        out[i] = f32_to_bf16(W[((long long)gate * WH + unit) * WH + k]);
    }
}

// bounded wait of ONE lane for *p >= want (relaxed agent-scope polls); false = gave up / someone else did
__device__ __forceinline__ bool wsr_wait_ge(gu32* p, unsigned want, gu32* gerr, unsigned* err, unsigned code) {
    unsigned spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(2);
        if (((++spins) & 1023u) == 0) {
            if (spins > (1u << 22) || __hip_atomic_load(gerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                atomicCAS(err, 0u, code);
                return false;
            }
        }
    }
    return true;
}

// ---- worker role (persistent launch): for every chunk a layer has finished AND normalised into the next
// layer's input rows (the recurrence CUs do the LayerNorm themselves, two rows each, from the h image
// they hold anyway), the next layer's input product G = X W_ih^T + b for the chunk, 128 x 256 tiles.
__device__ __forceinline__ void wsr_norm_row(const bf16_t* y0, const bf16_t* r0, const bf16_t* y1, const bf16_t* r1,
                                             const float* gamma, const float* beta, bf16_t* out, float scale, float eps,
                                             float* mean0, float* rstd0, float* mean1, float* rstd1, int lane) {
    // H = 1024: lane owns elements [16 lane, 16 lane + 16)
    float acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const bf16_t* y = k ? y1 : y0;
        const bf16_t* r = k ? r1 : r0;
        if (!y) continue;
        float a[16];
        ElemIO<bf16_t>::load_vec(y + lane * 16, *reinterpret_cast<float(*)[8]>(a));
        ElemIO<bf16_t>::load_vec(y + lane * 16 + 8, *reinterpret_cast<float(*)[8]>(a + 8));
        if (r) {
            float b[16];
            ElemIO<bf16_t>::load_vec(r + lane * 16, *reinterpret_cast<float(*)[8]>(b));
            ElemIO<bf16_t>::load_vec(r + lane * 16 + 8, *reinterpret_cast<float(*)[8]>(b + 8));
#pragma unroll
            for (int e = 0; e < 16; ++e) a[e] += b[e];
        }
        float sm = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) sm += a[e];
        const float mean = wave_sum(sm) * (1.f / WH);
        float sq = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) sq += (a[e] - mean) * (a[e] - mean);
        const float rstd = rsqrtf(wave_sum(sq) * (1.f / WH) + eps);
        if (lane == 0) {
            *(k ? mean1 : mean0) = mean;
            *(k ? rstd1 : rstd0) = rstd;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e)
            acc[e] += (a[e] - mean) * rstd * gamma[lane * 16 + e] + beta[lane * 16 + e];
    }
    float o[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = acc[e] * scale;
    ElemIO<bf16_t>::store_vec(out + lane * 16, *reinterpret_cast<float(*)[8]>(o));
    ElemIO<bf16_t>::store_vec(out + lane * 16 + 8, *reinterpret_cast<float(*)[8]>(o + 8));
}

// one 128 x 256 tile of C[M, 4H] = A[M, H] Bw[4H, H]^T + bias (bf16; K = H = 1024): 4 waves (2 x 2), wave
// tile 64 x 128 = 4 x 8 MFMA tiles.  ONE workgroup per CU here, and the operands come from MALL/HBM (40 MB
// of W_ih cycle through two 4 MB L2s): a K tile per memory round trip is latency-bound (measured 28 us
// per tile with one K tile in flight).  Three LDS stages of 48 KB filled by LDS-DMA, TWO K tiles in flight:
// `s_waitcnt vmcnt(12)` retires the oldest stage's 12 DMA instructions of this wave, a raw s_barrier
// publishes everybody's, then stage kt+2 is issued into the buffer all waves left two barriers ago.
// 128-byte LDS rows, 16-byte chunk index XOR-ed with row & 7 on the SOURCE address (gemm_nt.hip).
__device__ __forceinline__ void wsr_gemm_tile(const bf16_t* A, int M, const bf16_t* Bw, const float* bias, bf16_t* C,
                                              int m0, int n0, unsigned char* smem) {
    constexpr int BK = 64, A_BYTES = 128 * BK * 2, B_BYTES = 256 * BK * 2, BUFB = A_BYTES + B_BYTES;
    constexpr int KT = WH / BK, CCH = 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, r16 = lane & 15, kq = lane >> 4;
    const int prow = lane >> 3, chunk = (lane & 7) ^ prow;
    // pieces of 8 rows x 128 B: A has 16 (wave w: 4 i + w), B has 32 (wave w: 4 i + w, i < 8)
    auto issue = [&](int kt) {
        unsigned char* base = smem + (kt % 3) * BUFB;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = i * 4 + wave;
            const bf16_t* src = A + (long long)min(m0 + p * 8 + prow, M - 1) * WH + chunk * 8 + k0;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(base + p * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = i * 4 + wave;
            const bf16_t* src = Bw + (long long)(n0 + p * 8 + prow) * WH + chunk * 8 + k0;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(base + A_BYTES + p * 1024), 16, 0, 0);
        }
    };
    f32x4_t acc[4][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 bv = *reinterpret_cast<const float4*>(bias + n0 + wn * 128 + j * 16 + kq * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = (f32x4_t){bv.x, bv.y, bv.z, bv.w};
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");     // the previous tile's C staging reads are done (WAR on smem)
    issue(0);
    issue(1);
    for (int kt = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (kt + 2 < KT) issue(kt + 2);
        const unsigned char* sA = smem + (kt % 3) * BUFB;
        const unsigned char* sB = sA + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t a[4], b[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = wm * 64 + i * 16 + r16;
                a[i] = *reinterpret_cast<const bf16x8_t*>(sA + r * 128 + (((ks * 4 + kq) ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = wn * 128 + j * 16 + r16;
                b[j] = *reinterpret_cast<const bf16x8_t*>(sB + r * 128 + (((ks * 4 + kq) ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    unsigned char* sC = smem;        // [128 rows][32 chunks of 16 B], chunk ^= row & 31
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int nl = wn * 128 + j * 16 + kq * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ml = wm * 64 + i * 16 + r16;
            uint2 pk;
            pk.x = f32x2_to_bf16x2(acc[i][j][0], acc[i][j][1]);
            pk.y = f32x2_to_bf16x2(acc[i][j][2], acc[i][j][3]);
            *reinterpret_cast<uint2*>(sC + ml * 512 + ((((nl >> 3) ^ (ml & (CCH - 1))) << 4) | ((nl & 4) << 1))) = pk;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int c = threadIdx.x + it * 256;
        const int rl = c / CCH, ch = c % CCH;
        if (m0 + rl >= M) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(sC + rl * 512 + ((ch ^ (rl & (CCH - 1))) << 4));
        *reinterpret_cast<uint4*>(C + (long long)(m0 + rl) * 4 * WH + n0 + ch * 8) = v;
    }
}

__device__ void wsr_worker(const EdWsrLaunch& L, int wi, int NW, unsigned char* smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int B = L.B;
    const long long BH = (long long)B * WH;
    gu32* gerr = (gu32*)L.err;
    __shared__ unsigned ok_s;
    int nch = 0;
    for (int l = 0; l < L.nslot; ++l) nch = max(nch, (L.slot[l].T + L.slot[l].cf - 1) / L.slot[l].cf);
    // wavefront order: layer l's chunk k is finished about l chunk-times after layer 0's
    for (int w = 0; w < nch + L.nslot; ++w) {
        for (int l = 0; l < L.nslot; ++l) {
            const int k = w - l;
            const EdWsrSlot& S = L.slot[l];
            if (k < 0 || k * S.cf >= S.T) continue;
            const int t0 = k * S.cf, t1 = min(S.T, t0 + S.cf);
            long long* tr = (L.trace && wi == 0 && threadIdx.x == 0 && k < 64) ? L.trace + 2048 + (l * 64 + k) * 4 : nullptr;
            if (tr) tr[0] = wall_clock64();
            // ---- frames [t0, t1) of layer l are out AND normalised (next layer's input rows written)
            if (threadIdx.x == 0)
                ok_s = wsr_wait_ge((gu32*)S.ydone, (unsigned)(WCUS * (k + 1)), gerr, L.err, 300u + l) ? 1u : 0u;
            __syncthreads();
            if (!ok_s) return;
            if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __syncthreads();
            if (tr) tr[1] = wall_clock64();
            const int r = S.reduce, tau0 = t0 / r, ntau = (t1 - t0 + r - 1) / r;
            if (!WSR_LN_IN_RECURRENCE) {
                // ---- LayerNorm rows: output frames tau in [tau0, tau0 + ntau), one wave per (tau, b)
                for (int row = wi * 4 + wave; row < ntau * B; row += NW * 4) {
                    const int tau = tau0 + row / B, b = row % B;
                    const int ta = tau * r, tb = (r == 2 && ta + 1 < S.T) ? ta + 1 : -1;
                    const bf16_t* y0 = S.Y + (long long)(ta - S.t0) * BH + (long long)b * WH;
                    const bf16_t* y1 = tb >= 0 ? S.Y + (long long)(tb - S.t0) * BH + (long long)b * WH : nullptr;
                    const bf16_t* r0 = S.X ? S.X + (long long)ta * BH + (long long)b * WH : nullptr;
                    const bf16_t* r1 = (tb >= 0 && S.X) ? S.X + (long long)tb * BH + (long long)b * WH : nullptr;
                    wsr_norm_row(y0, r0, y1, r1, S.gamma, S.beta, S.nX + (long long)tau * S.nX_st + (long long)b * S.nX_sb,
                                 r == 1 ? 1.f : 0.5f, L.eps, S.mean + (long long)ta * B + b, S.rstd + (long long)ta * B + b,
                                 tb >= 0 ? S.mean + (long long)tb * B + b : nullptr,
                                 tb >= 0 ? S.rstd + (long long)tb * B + b : nullptr, lane);
                }
                if (!S.nWih) continue;      // last layer: the rows above ARE the stack output
                // publish the rows, wait for every worker's
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_fetch_add((gu32*)(S.xdone + k), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok_s = wsr_wait_ge((gu32*)(S.xdone + k), (unsigned)NW, gerr, L.err, 400u + l) ? 1u : 0u;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                if (!ok_s) return;
            }
            if (!S.nWih) continue;          // last layer: its LayerNorm rows ARE the stack output
            if (tr) tr[2] = wall_clock64();
            // ---- input product of layer l + 1 for these rows: tiles mt x 32
            const int M = ntau * B;
            const bf16_t* A = S.nX + (long long)tau0 * S.nX_st;          // [M, H] (time-major rows: nX_st = B H)
            bf16_t* C = S.nG + (long long)tau0 * B * 4 * WH;
            const int mtiles = (M + 127) / 128;
            for (int tile = wi; tile < mtiles * 16; tile += NW)
                wsr_gemm_tile(A, M, S.nWih, S.nBias, C, (tile / 16) * 128, (tile % 16) * 256, smem);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add((gu32*)(S.ngdone + k), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tr) tr[3] = wall_clock64();
        }
    }
}

// LayerNorm of ONE row of frame t of a layer, by one wave of a recurrence CU, from the h image that CU
// holds in LDS (A-fragment order) + the residual row; under time reduction the even frame's row waits in
// `prev` (registers) for its partner.  Writes the next layer's input row (or the stack output) with
// write-through stores: the workers on another XCD read it inside this launch.
__device__ __forceinline__ void wsr_ln_from_image(const EdWsrSlot& S, const unsigned char* img_lds, int t, int b, int B,
                                                  float eps, unsigned (&prev)[8], int lane) {
    const int rbk = b >> 4, r16 = b & 15;
    float a[16];
    {
        const unsigned char* p0 = img_lds + ((lane >> 1) * 4 + rbk) * 1024 + (((lane & 1) * 2) * 16 + r16) * 16;
        ElemIO<bf16_t>::load_vec(reinterpret_cast<const bf16_t*>(p0), *reinterpret_cast<float(*)[8]>(a));
        ElemIO<bf16_t>::load_vec(reinterpret_cast<const bf16_t*>(p0 + 256), *reinterpret_cast<float(*)[8]>(a + 8));
    }
    if (S.X) {
        const bf16_t* r = S.X + ((long long)t * B + b) * WH + lane * 16;
        float x[16];
        ElemIO<bf16_t>::load_vec(r, *reinterpret_cast<float(*)[8]>(x));
        ElemIO<bf16_t>::load_vec(r + 8, *reinterpret_cast<float(*)[8]>(x + 8));
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] += x[e];
    }
    float sm = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) sm += a[e];
    const float mean = wave_sum(sm) * (1.f / WH);
    float sq = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) sq += (a[e] - mean) * (a[e] - mean);
    const float rstd = rsqrtf(wave_sum(sq) * (1.f / WH) + eps);
    if (lane == 0) {
        S.mean[(long long)t * B + b] = mean;
        S.rstd[(long long)t * B + b] = rstd;
    }
    float o[16];
#pragma unroll
    for (int e = 0; e < 16; e += 4) {
        const float4 gm = *reinterpret_cast<const float4*>(S.gamma + lane * 16 + e);
        const float4 bt = *reinterpret_cast<const float4*>(S.beta + lane * 16 + e);
        o[e] = (a[e] - mean) * rstd * gm.x + bt.x;
        o[e + 1] = (a[e + 1] - mean) * rstd * gm.y + bt.y;
        o[e + 2] = (a[e + 2] - mean) * rstd * gm.z + bt.z;
        o[e + 3] = (a[e + 3] - mean) * rstd * gm.w + bt.w;
    }
    bool emit = true;
    int tau = t;
    if (S.reduce == 2) {
        tau = t >> 1;
        if (!(t & 1)) {
            if (t + 1 < S.T) {       // wait for the partner frame (kept as packed bf16: 8 registers)
#pragma unroll
                for (int e = 0; e < 8; ++e) prev[e] = f32x2_to_bf16x2(o[2 * e], o[2 * e + 1]);
                emit = false;
            } else {                 // odd length: the zero-padded partner counts as 0
#pragma unroll
                for (int e = 0; e < 16; ++e) o[e] *= 0.5f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[2 * e] = 0.5f * (__uint_as_float(prev[e] << 16) + o[2 * e]);
                o[2 * e + 1] = 0.5f * (__uint_as_float(prev[e] & 0xffff0000u) + o[2 * e + 1]);
            }
        }
    }
    if (emit) {
        u32x4_t v0, v1;
        v0[0] = f32x2_to_bf16x2(o[0], o[1]);   v0[1] = f32x2_to_bf16x2(o[2], o[3]);
        v0[2] = f32x2_to_bf16x2(o[4], o[5]);   v0[3] = f32x2_to_bf16x2(o[6], o[7]);
        v1[0] = f32x2_to_bf16x2(o[8], o[9]);   v1[1] = f32x2_to_bf16x2(o[10], o[11]);
        v1[2] = f32x2_to_bf16x2(o[12], o[13]); v1[3] = f32x2_to_bf16x2(o[14], o[15]);
        bf16_t* dst = S.nX + (long long)tau * S.nX_st + (long long)b * S.nX_sb + lane * 16;
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v0) : "memory");
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst + 8), "v"(v1) : "memory");
    }
}

__global__ __launch_bounds__(256, 1) void wsr_fwd_kernel(EdWsrLaunch L) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char* gtile = lds + W_IMG;                       // [64 rows][128 cols] bf16
    unsigned char* hstage = gtile + W_GT;                     // [64 rows][32 units] bf16
    float* cstage = reinterpret_cast<float*>(hstage + W_HS);  // [64 rows][32 units] f32
    __shared__ unsigned role_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    if ((int)xcc >= L.nslot && !L.persistent) return;   // no layer for this XCD in this launch
    if (threadIdx.x == 0) role_s = atomicAdd(&L.ticket[xcc], 1u);
    __syncthreads();
    const int cu = (int)role_s;
    if (cu >= WCUS) {                            // more than 32 workgroups of this grid landed on the XCD
        if (threadIdx.x == 0) atomicExch(L.err, 7000u + xcc);
        return;
    }
    if ((int)xcc >= L.nslot) {                   // persistent launch: the spare XCDs' workgroups are workers
        wsr_worker(L, ((int)xcc - L.nslot) * WCUS + cu, (8 - L.nslot) * WCUS, lds);
        return;
    }
    const EdWsrSlot& S = L.slot[xcc];
    const int B = L.B, MT = (B + 15) >> 4;
    const long long BH = (long long)B * WH;
    gu32* cnt = (gu32*)S.counter;
    gu32* gerr = (gu32*)L.err;

    // ---- stationary weights: 2 tiles x 32 k-steps x 8 bf16 = 256 registers per lane
    bf16x8_t w[2][WKS];
    {
        const bf16x8_t* src = reinterpret_cast<const bf16x8_t*>(S.Wreg) +
                              (((long long)cu * 4 + wave) * 2 * WKS) * 64 + lane;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int ks = 0; ks < WKS; ++ks) w[tn][ks] = src[(tn * WKS + ks) * 64];
    }
    const int n = lane & 15, q = lane >> 4;
    const bool lo = n < 8;                        // lo lanes finish rows q*4 + {0,1}, hi lanes rows {2,3}
    const int jl = wave * 8 + (n & 7);            // unit within this CU's 32
    const int gcol = (jl >> 4) * 64 + (jl & 15);  // + gate * 16: column within the CU's 128 (ed_gate_col)

    // ---- cell state of this lane's 8 (row, unit) cells, from c_{t0-1}
    float c[4][2];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = rb * 16 + q * 4 + (lo ? 0 : 2) + r;
            c[rb][r] = row < B ? S.C_prev[(long long)row * WH + cu * WUPC + jl] : 0.f;
        }

    // G slice of a frame: rows x 256 bytes at column 128 cu; 16 lanes cover one row
    const int grow = threadIdx.x >> 4, gch = threadIdx.x & 15;
    auto g_ptr = [&](int s, int i) {
        return reinterpret_cast<u32x4_t*>(S.G + ((long long)s * B + grow + 16 * i) * 4 * WH + cu * 128 + gch * 8);
    };
    u32x4_t gq[4];
    unsigned ln_prev[8];     // LayerNorm row of an even frame (packed bf16), waiting for its time-reduction partner
#pragma unroll
    for (int e = 0; e < 8; ++e) ln_prev[e] = 0u;
    const int NW = (8 - L.nslot) * WCUS;
    if (!(L.persistent && S.gdone)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) gq[i] = (grow + 16 * i < B) ? *g_ptr(0, i) : (u32x4_t){0u, 0u, 0u, 0u};
    }

    for (int s = 0; s < S.nsteps; ++s) {
        const int t = S.t0 + s;
        if (L.trace && cu == 0 && threadIdx.x == 0 && (t % S.cf) == 0 && t / S.cf < 64)
            L.trace[(xcc * 64 + t / S.cf) * 4 + 0] = wall_clock64();
        if (L.persistent && S.gdone && (t % S.cf) == 0) {
            // a new chunk of this layer's input product: wait until every worker has delivered its tiles
            if (threadIdx.x == 0) {
                role_s = wsr_wait_ge((gu32*)(S.gdone + t / S.cf), (unsigned)NW, gerr, L.err, 200u + xcc) ? 1u : 0u;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            if (!role_s) return;
#pragma unroll
            for (int i = 0; i < 4; ++i) gq[i] = (grow + 16 * i < B) ? *g_ptr(s, i) : (u32x4_t){0u, 0u, 0u, 0u};
        }
        if (L.trace && cu == 0 && threadIdx.x == 0 && (t % S.cf) == 0 && t / S.cf < 64)
            L.trace[(xcc * 64 + t / S.cf) * 4 + 1] = wall_clock64();
        // pre-activations of this step -> LDS tile (the previous step's gate store has read it: the
        // barrier of the wait below orders this write before any lane's read of the new contents)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<u32x4_t*>(gtile + ((grow + 16 * i) * 128 + gch * 8) * 2) = gq[i];
        // ---- wait for h_{t-1} of every CU of this layer
        if (threadIdx.x == 0) {
            const unsigned want = S.base + (unsigned)(WCUS * s);
            unsigned spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (((++spins) & 1023u) == 0) {
                    if (spins > (1u << 21) ||
                        __hip_atomic_load(gerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                        atomicCAS(L.err, 0u, 100u + xcc);
                        break;
                    }
                }
            }
        }
        __syncthreads();
        if (__hip_atomic_load(gerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
        // ---- gather the image of h_{t-1}: MT x 32 pieces of 1 KB, L2-served (sc1: never a stale L1 line);
        // wave w brings row block w of every k-step
        {
            const unsigned char* img = reinterpret_cast<const unsigned char*>((t & 1) ? S.img1 : S.img0);
            if (wave < MT) {
#pragma unroll
                for (int ks = 0; ks < WKS; ++ks)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(img + ((long long)(ks * MT + wave) * 64 + lane) * 16),
                        (__attribute__((address_space(3))) void*)(lds + (ks * 4 + wave) * 1024), 16, 0, 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // ---- gates = G_t + h_{t-1} W^T; A fragments prefetched two k-steps ahead
        f32x4_t acc[4][2];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            acc[rb][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            acc[rb][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
        {
            bf16x8_t af[3][4];
            auto lda = [&](int ks, int slot) {
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
                    af[slot][rb] = *reinterpret_cast<const bf16x8_t*>(lds + (ks * 4 + rb) * 1024 + lane * 16);
            };
            lda(0, 0);
            lda(1, 1);
#pragma unroll
            for (int ks = 0; ks < WKS; ++ks) {
                if (ks + 2 < WKS) lda(ks + 2, (ks + 2) % 3);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks % 3][rb], w[0][ks], acc[rb][0], 0, 0, 0);
                    acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks % 3][rb], w[1][ks], acc[rb][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- cell update.  Tile 0 columns = [i(8 units) | f(8 units)], tile 1 = [g | o]; lane (n, q) holds
        // rows q*4 + r.  Lanes n and n^8 swap halves (DPP row_ror:8): lo lanes finish rows r = 0,1, hi lanes
        // rows r = 2,3.  Rows of row blocks >= MT read zero pre-activations and are never stored.
        bf16_t* gt16 = reinterpret_cast<bf16_t*>(gtile);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            float mine0[2], mine1[2], send0[2], send1[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                mine0[r] = lo ? acc[rb][0][r] : acc[rb][0][2 + r];
                mine1[r] = lo ? acc[rb][1][r] : acc[rb][1][2 + r];
                send0[r] = lo ? acc[rb][0][2 + r] : acc[rb][0][r];
                send1[r] = lo ? acc[rb][1][2 + r] : acc[rb][1][r];
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float o0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send0[r]), 0x128, 0xf, 0xf, false));
                const float o1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send1[r]), 0x128, 0xf, 0xf, false));
                const float pi = lo ? mine0[r] : o0, pf = lo ? o0 : mine0[r];
                const float pg = lo ? mine1[r] : o1, po = lo ? o1 : mine1[r];
                const int row = rb * 16 + q * 4 + (lo ? 0 : 2) + r;
                bf16_t* gp = gt16 + row * 128 + gcol;
                const float gi = wsigm(pi + bf16_to_f32(gp[0]));
                const float gf = wsigm(pf + bf16_to_f32(gp[16]));
                const float gg = wtanh(pg + bf16_to_f32(gp[32]));
                const float go = wsigm(po + bf16_to_f32(gp[48]));
                gp[0] = f32_to_bf16(gi);           // the backward pass reads the GATES from G
                gp[16] = f32_to_bf16(gf);
                gp[32] = f32_to_bf16(gg);
                gp[48] = f32_to_bf16(go);
                const float cn = gf * c[rb][r] + gi * gg;
                c[rb][r] = cn;
                const float hn = go * wtanh(cn);
                reinterpret_cast<bf16_t*>(hstage)[row * WUPC + jl] = (row < B) ? f32_to_bf16(hn) : (bf16_t)0;
                cstage[row * WUPC + jl] = cn;
            }
        }
        __syncthreads();
        // ---- publish FIRST: this CU's slice of the next image = k-step `cu`, row block = wave: one 16-byte
        // write-through store per lane; drain; arrive
        if (wave < MT) {
            const int m = wave * 16 + (lane & 15), k0 = (lane >> 4) * 8;
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(hstage + (m * WUPC + k0) * 2);
            bf16_t* dst = ((t & 1) ? S.img0 : S.img1) + ((long long)(cu * MT + wave) * 64 + lane) * 8;
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // EVERY storing wave drains
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- off the critical path: Y_t, c_t rows and the gates (plain 16-byte stores), next frame's
        // pre-activations (into registers; written to the LDS tile at the top of the next iteration)
        {
            const int row = threadIdx.x >> 2, ch = threadIdx.x & 3;
            if (row < B) {
                const u32x4_t y = *reinterpret_cast<const u32x4_t*>(hstage + (row * WUPC + ch * 8) * 2);
                bf16_t* yd = S.Y + (long long)s * BH + (long long)row * WH + cu * WUPC + ch * 8;
                if (L.persistent && !WSR_LN_IN_RECURRENCE)   // read by the workers inside this launch: write-through
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(yd), "v"(y) : "memory");
                else
                    *reinterpret_cast<u32x4_t*>(yd) = y;
                const uint4 c0 = *reinterpret_cast<const uint4*>(cstage + row * WUPC + ch * 8);
                const uint4 c1 = *reinterpret_cast<const uint4*>(cstage + row * WUPC + ch * 8 + 4);
                float* cd = S.C + (long long)s * BH + (long long)row * WH + cu * WUPC + ch * 8;
                *reinterpret_cast<uint4*>(cd) = c0;
                *reinterpret_cast<uint4*>(cd + 4) = c1;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (grow + 16 * i < B)
                    *g_ptr(s, i) = *reinterpret_cast<const u32x4_t*>(gtile + ((grow + 16 * i) * 128 + gch * 8) * 2);
                if (s + 1 < S.nsteps && !(L.persistent && S.gdone && ((t + 1) % S.cf) == 0))
                    gq[i] = (grow + 16 * i < B) ? *g_ptr(s + 1, i) : (u32x4_t){0u, 0u, 0u, 0u};
            }
        }
        if (L.persistent && !WSR_LN_IN_RECURRENCE && (((t + 1) % S.cf) == 0 || s + 1 == S.nsteps)) {
            // a chunk of this layer is complete: its h rows (write-through stores) are out once every wave
            // has drained; tell the workers
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0)
                __hip_atomic_fetch_add((gu32*)S.ydone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (L.trace && cu == 0 && threadIdx.x == 0 && t / S.cf < 64)
                L.trace[(xcc * 64 + t / S.cf) * 4 + 2] = wall_clock64();
        }
        if (L.persistent && WSR_LN_IN_RECURRENCE && t > 0) {
            // LayerNorm of frame t-1 (its h image is the one this step gathered): rows 2 cu and 2 cu + 1, one
            // wave each - after the publish, while the other CUs of the layer are still arriving
            if (wave < 2 && 2 * cu + wave < B) wsr_ln_from_image(S, lds, t - 1, 2 * cu + wave, B, L.eps, ln_prev, lane);
            if ((t % S.cf) == 0) {
                // ... which completes chunk t / cf - 1 of the next layer's input: tell the workers
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0)
                    __hip_atomic_fetch_add((gu32*)S.ydone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (L.trace && cu == 0 && threadIdx.x == 0 && t / S.cf - 1 < 64)
                    L.trace[(xcc * 64 + t / S.cf - 1) * 4 + 2] = wall_clock64();
            }
        }
        __syncthreads();      // the LDS tile is rewritten at the top of the next iteration
    }
    if (L.persistent && WSR_LN_IN_RECURRENCE) {
        // the last frame: wait for its image, normalise, complete the last chunk
        const int t = S.t0 + S.nsteps;
        if (threadIdx.x == 0)
            role_s = wsr_wait_ge(cnt, S.base + (unsigned)(WCUS * S.nsteps), gerr, L.err, 500u + xcc) ? 1u : 0u;
        __syncthreads();
        if (!role_s) return;
        const unsigned char* img = reinterpret_cast<const unsigned char*>((t & 1) ? S.img1 : S.img0);
        if (wave < MT) {
#pragma unroll
            for (int ks = 0; ks < WKS; ++ks)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(img + ((long long)(ks * MT + wave) * 64 + lane) * 16),
                    (__attribute__((address_space(3))) void*)(lds + (ks * 4 + wave) * 1024), 16, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (wave < 2 && 2 * cu + wave < B) wsr_ln_from_image(S, lds, t - 1, 2 * cu + wave, B, L.eps, ln_prev, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_fetch_add((gu32*)S.ydone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (L.trace && cu == 0 && threadIdx.x == 0 && (t - 1) / S.cf < 64)
            L.trace[(xcc * 64 + (t - 1) / S.cf) * 4 + 2] = wall_clock64();
    }
}

}  // namespace

int ed_wsr_pack_fwd(const float* w_hh, bf16_t* out, hipStream_t s) {
    hipLaunchKernelGGL(wsr_pack_fwd_kernel, dim3(4096), dim3(256), 0, s, w_hh, out);
    ED_CHECK_LAUNCH("wsr_pack_fwd_kernel");
    return ED_OK;
}

int ed_wsr_workers(const EdWsrLaunch& L) { return (8 - L.nslot) * WCUS; }

int ed_wsr_launch_fwd(const EdWsrLaunch& L, hipStream_t s) {
    if (L.nslot == 0) return ED_OK;
    // (per device; cheap enough to repeat: the caller holds the stack's mutex)
    ED_CHECK_HIP(hipFuncSetAttribute((const void*)wsr_fwd_kernel,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS));
    hipLaunchKernelGGL(wsr_fwd_kernel, dim3(256), dim3(256), W_LDS, s, L);
    ED_CHECK_LAUNCH("wsr_fwd_kernel");
    return ED_OK;
}
