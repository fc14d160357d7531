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
    if ((int)xcc >= L.nslot) return;             // no layer for this XCD in this launch
    if (threadIdx.x == 0) role_s = atomicAdd(&L.ticket[xcc], 1u);
    __syncthreads();
    const int cu = (int)role_s;
    if (cu >= WCUS) {                            // more than 32 workgroups of this grid landed on the XCD
        if (threadIdx.x == 0) atomicExch(L.err, 7000u + xcc);
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
#pragma unroll
    for (int i = 0; i < 4; ++i) gq[i] = (grow + 16 * i < B) ? *g_ptr(0, i) : (u32x4_t){0u, 0u, 0u, 0u};

    for (int s = 0; s < S.nsteps; ++s) {
        const int t = S.t0 + s;
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
                const uint4 y = *reinterpret_cast<const uint4*>(hstage + (row * WUPC + ch * 8) * 2);
                *reinterpret_cast<uint4*>(S.Y + (long long)s * BH + (long long)row * WH + cu * WUPC + ch * 8) = y;
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
                if (s + 1 < S.nsteps)
                    gq[i] = (grow + 16 * i < B) ? *g_ptr(s + 1, i) : (u32x4_t){0u, 0u, 0u, 0u};
            }
        }
        __syncthreads();      // the LDS tile is rewritten at the top of the next iteration
    }
}

}  // namespace

int ed_wsr_pack_fwd(const float* w_hh, bf16_t* out, hipStream_t s) {
    hipLaunchKernelGGL(wsr_pack_fwd_kernel, dim3(4096), dim3(256), 0, s, w_hh, out);
    ED_CHECK_LAUNCH("wsr_pack_fwd_kernel");
    return ED_OK;
}

int ed_wsr_launch_fwd(const EdWsrLaunch& L, hipStream_t s) {
    if (L.nslot == 0) return ED_OK;
    // (per device; cheap enough to repeat: the caller holds the stack's mutex)
    ED_CHECK_HIP(hipFuncSetAttribute((const void*)wsr_fwd_kernel,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS));
    hipLaunchKernelGGL(wsr_fwd_kernel, dim3(256), dim3(256), W_LDS, s, L);
    ED_CHECK_LAUNCH("wsr_fwd_kernel");
    return ED_OK;
}
