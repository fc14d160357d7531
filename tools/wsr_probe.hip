// Probe: WEIGHTS-STATIONARY LSTM recurrence, one layer per XCD, h exchanged through that XCD's L2.
//
// Question ("next round" item 8): what does ONE time step of the [64,1024] x [1024,4096]
// recurrent product cost when W_hh never leaves the chip - 32 workgroups (one per CU of an XCD), each
// keeping its 128 gate columns x 1024 of W_hh (256 KB bf16) in its REGISTER FILE (4 waves x 256
// VGPRs), h_t (128 KB bf16) all-gathered per step through L2 with a monotonic arrival counter -
// against today's launch-per-step design (15.8 us per 4-layer-step forward launch, W_hh re-streamed
// from MALL/HBM every step)?
//
// The kernel below is the real forward step of one layer (rnnt/models.py:65 -> torch LSTM cell,
// gate order i,f,g,o), for 8 independent layer instances (one per XCD) so that the whole chip is
// loaded the way the real encoder would load it:
//   per step and workgroup:  poll counter >= 32 t  ->  LDS-DMA gather of the 128 KB h image
//   (fragment order, sc1 = L2-served)  ->  256 MFMA 16x16x32 per wave, B operand = registers  ->
//   cell update in registers (c_t stays in registers)  ->  h tile -> LDS -> 16-byte sc1 stores of
//   this CU's 4 KB slice of the next image + plain stores of Y_t, c_t  ->  counter += 1.
// Every spin is bounded (give-up code in err[0]); results are checked against a CPU recurrence.
//
//   hipcc -O3 --offload-arch=gfx950 tools/wsr_probe.hip -o tools/wsr_probe.bin
//   tools/wsr_probe.bin [T=400] [mode]      mode bits: 1 skip MFMA, 2 skip exchange, 4 one instance, 8 gather through registers,
//                                        16 plain loads + agent acquire fence, 32 plain (write-back) image stores,
//                                        64 whole-image gather before the first MFMA (no chunk overlap),
//                                        128 spread every layer over all 8 XCDs (4 CUs each): cross-XCD exchange
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));            \
            exit(2);                                                                  \
        }                                                                             \
    } while (0)

constexpr int B = 64, H = 1024, CUS = 32, UPC = H / CUS;   // 32 units per CU
constexpr int KS = H / 32;                                  // 32 k-steps of 32
constexpr int IMG_BYTES = B * H * 2;                        // 128 KB h image (fragment order)
constexpr int NINST = 8;

struct Args {
    const bf16_t* Wreg;   // [inst][cu][wave][tile 2][ks 32][lane 64][8]
    const bf16_t* G;      // [inst][T][cu][wave][lane][32]   pre-activations, this lane's 8 cells x 4 gates
    bf16_t* himg;         // [inst][2][B*H] fragment-order images (ping-pong); image 0 zero = h_{-1}
    bf16_t* Y;            // [inst][T][B][H]
    float* C;             // [inst][T][B][H]
    unsigned* counter;    // [inst] arrivals
    unsigned* ticket;     // [8] per-XCD role tickets
    unsigned* err;        // [0] give-up code, [1..8] census
    long long* cycles;    // [inst] wall clock of cu 0
    long long* phase;     // [8] summed wall-clock ticks of the phases of (inst 0, cu 0, thread 0)
    int T, mode;
};

__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((unsigned)b) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    const __bf16 b = (__bf16)f;
    return *reinterpret_cast<const bf16_t*>(&b);
}
// v_exp_f32 / v_rcp_f32 (1 ulp) - the results are rounded to bf16 anyway
__device__ __forceinline__ float sigm(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanh_(float x) {
    const float xc = fminf(fmaxf(x, -15.f), 15.f);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * xc));
}

__device__ __forceinline__ uint4 ld_g(const bf16_t* G, int T, int inst, int t, int cu, int wave, int lane, int i) {
    const uint4* gsrc = reinterpret_cast<const uint4*>(
        G + ((((long long)inst * T + t) * CUS + cu) * 4 + wave) * 64 * 32 + lane * 32);
    return gsrc[i];
}

__global__ __launch_bounds__(256, 1) void wsr_probe(Args a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];   // image | h stage 4 KB | c stage 8 KB
    unsigned char* hstage = lds + IMG_BYTES;
    float* cstage = reinterpret_cast<float*>(lds + IMG_BYTES + 4096);
    __shared__ unsigned role_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    if (threadIdx.x == 0) {
        role_s = atomicAdd(&a.ticket[xcc], 1u);
        atomicAdd(&a.err[1 + xcc], 1u);
    }
    __syncthreads();
    int cu = (int)role_s;
    int inst = (int)xcc;
    if (cu >= CUS) return;                       // surplus block on this XCD (census reports it)
    if (a.mode & 128) {                          // SPREAD: every instance takes 4 CUs of every XCD
        inst = cu >> 2;
        cu = (int)xcc * 4 + (cu & 3);
    }
    if ((a.mode & 4) && inst != 0) return;

    // ---- stationary weights: 2 tiles x 32 k-steps x 8 bf16 = 256 VGPRs
    bf16x8_t w[2][KS];
    {
        const bf16x8_t* src = reinterpret_cast<const bf16x8_t*>(a.Wreg) +
                              ((((long long)inst * CUS + cu) * 4 + wave) * 2 * KS) * 64 + lane;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) w[tn][ks] = src[(tn * KS + ks) * 64];
    }
    float c[4][2];                                // cell state of this lane's 8 (row, unit) cells
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i][0] = c[i][1] = 0.f;

    const int n = lane & 15, q = lane >> 4;
    const bool lo = n < 8;                        // lo lanes finish rows q*4 + {0,1}, hi lanes rows {2,3}
    gu32* cnt = (gu32*)(a.counter + inst);
    const long long t_begin = wall_clock64();
    // buffer resource over the two images of this instance (register-gather arm)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.himg + (long long)inst * 2 * B * H), 0, 2 * IMG_BYTES, 0x00020000);

    uint4 gq0 = ld_g(a.G, a.T, inst, 0, cu, wave, lane, 0), gq1 = ld_g(a.G, a.T, inst, 0, cu, wave, lane, 1),
          gq2 = ld_g(a.G, a.T, inst, 0, cu, wave, lane, 2), gq3 = ld_g(a.G, a.T, inst, 0, cu, wave, lane, 3);

    long long ph[7] = {0, 0, 0, 0, 0, 0, 0};
#define STAMP(i)                                   \
    do {                                           \
        const long long now_ = wall_clock64();     \
        ph[i] += now_ - last_;                     \
        last_ = now_;                              \
    } while (0)
    for (int t = 0; t < a.T; ++t) {
        long long last_ = wall_clock64();
        // ---- wait for h_{t-1} of every CU of this layer
        if (!(a.mode & 2)) {
            if (threadIdx.x == 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(CUS * t)) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) {
                        atomicExch(&a.err[0], 100u + (unsigned)inst);
                        break;
                    }
                }
                if (a.mode & 16) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            STAMP(0);
            if (__hip_atomic_load((gu32*)a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
            // ---- gather the image: 128 x 1 KB pieces, 32 per wave, L2-served (sc1: never the stale L1)
            if (a.mode & 8) {
                // through registers: 4 batches of 8 x 16-byte sc1 buffer loads, then ds_write_b128
#pragma unroll
                for (int bt = 0; bt < 4; ++bt) {
                    u32x4_t v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int piece = (bt * 8 + i) * 4 + wave;
                        v[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (t & 1) * IMG_BYTES + piece * 1024 + lane * 16, 0, 16);
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int piece = (bt * 8 + i) * 4 + wave;
                        *reinterpret_cast<u32x4_t*>(lds + piece * 1024 + lane * 16) = v[i];
                    }
                }
            } else {
                // LDS-DMA, issued in k order: piece = (ks*4 + rb); wave w issues the pieces of k-steps
                // {4j + w}: after its first 8 issues every wave has covered k-steps 0..7 x 4 row blocks?
                // no - piece index p*4 + wave walks (ks = p, rb = wave): wave w brings row block w of
                // every k-step, in k order, so "the first 8 of each wave" = k-steps 0..7 complete.
                const unsigned char* img = reinterpret_cast<const unsigned char*>(
                    a.himg + ((long long)inst * 2 + (t & 1)) * B * H);
#pragma unroll
                for (int p = 0; p < 32; ++p) {
                    const int piece = p * 4 + wave;
                    if (a.mode & 16)
                        __builtin_amdgcn_global_load_lds(
                            (const __attribute__((address_space(1))) void*)(img + piece * 1024 + lane * 16),
                            (__attribute__((address_space(3))) void*)(lds + piece * 1024), 16, 0, 0);
                    else
                        __builtin_amdgcn_global_load_lds(
                            (const __attribute__((address_space(1))) void*)(img + piece * 1024 + lane * 16),
                            (__attribute__((address_space(3))) void*)(lds + piece * 1024), 16, 0, 16);
                }
                if (a.mode & 64) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                }
            }
            if (a.mode & 8) __syncthreads();
            STAMP(1);
        }

        // ---- gates = G_t + h_{t-1} W^T : 4 row blocks x 2 column tiles; A fragments prefetched two
        // k-steps ahead; with the chunked gather (default) the MFMAs of k-steps 8c..8c+7 start as soon
        // as chunk c (32 KB) of the image has landed for EVERY wave (counted vmcnt + barrier).
        f32x4_t acc[4][2];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            acc[rb][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            acc[rb][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
        if (!(a.mode & 1)) {
            const bool chunked = !(a.mode & (2 | 8 | 64));
            bf16x8_t af[3][4];
            auto lda = [&](int ks, int slot) {
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
                    af[slot][rb] = *reinterpret_cast<const bf16x8_t*>(lds + (ks * 4 + rb) * 1024 + lane * 16);
            };
            if (chunked) {
                asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                __syncthreads();
            }
            lda(0, 0);
            lda(1, 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (chunked && (ks & 7) == 6 && ks + 2 < KS) {       // the prefetch below crosses into the next chunk
                    if (ks == 6) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    if (ks == 14) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    if (ks == 22) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                }
                if (ks + 2 < KS) lda(ks + 2, (ks + 2) % 3);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks % 3][rb], w[0][ks], acc[rb][0], 0, 0, 0);
                    acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks % 3][rb], w[1][ks], acc[rb][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("" ::"v"(acc[0][0]), "v"(acc[3][1]));
        STAMP(2);
        // ---- cell update.  Tile 0 columns = [i(8 units) | f(8 units)], tile 1 = [g | o]; lane (n, q)
        // holds rows q*4 + r.  Lanes n and n^8 swap halves: lo lanes end up with i,f,g,o of rows r = 0,1,
        // hi lanes of rows r = 2,3.  pre-activations: cell = rb*2 + r holds 4 bf16 (i,f,g,o) = 2 dwords
        const unsigned gw[16] = {gq0.x, gq0.y, gq0.z, gq0.w, gq1.x, gq1.y, gq1.z, gq1.w,
                                 gq2.x, gq2.y, gq2.z, gq2.w, gq3.x, gq3.y, gq3.z, gq3.w};
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
                // lanes n and n^8 of one 16-lane row: DPP row_ror:8 (VALU, no LDS round trip)
                const float o0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send0[r]), 0x128, 0xf, 0xf, false));
                const float o1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send1[r]), 0x128, 0xf, 0xf, false));
                const float pi = lo ? mine0[r] : o0, pf = lo ? o0 : mine0[r];
                const float pg = lo ? mine1[r] : o1, po = lo ? o1 : mine1[r];
                const int cell = rb * 2 + r;
                const float gi = sigm(pi + __uint_as_float(gw[cell * 2] << 16));
                const float gf = sigm(pf + __uint_as_float(gw[cell * 2] & 0xffff0000u));
                const float gg = tanh_(pg + __uint_as_float(gw[cell * 2 + 1] << 16));
                const float go = sigm(po + __uint_as_float(gw[cell * 2 + 1] & 0xffff0000u));
                const float cn = gf * c[rb][r] + gi * gg;
                c[rb][r] = cn;
                const float hn = go * tanh_(cn);
                const int row = rb * 16 + q * 4 + (lo ? 0 : 2) + r;
                reinterpret_cast<bf16_t*>(hstage)[row * UPC + wave * 8 + (n & 7)] = f2bf(hn);
                cstage[row * UPC + wave * 8 + (n & 7)] = cn;
            }
        }
        __syncthreads();
        STAMP(3);
        // ---- publish FIRST: this CU's slice of the next image = k-step `cu`, 4 row blocks = 4 KB, one
        // 16-byte write-through store per lane (wave = row block); drain; arrive.
        if (!(a.mode & 2)) {
            const int m = wave * 16 + (lane & 15), k0 = (lane >> 4) * 8;
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(hstage + (m * UPC + k0) * 2);
            bf16_t* dst = a.himg + ((long long)inst * 2 + ((t + 1) & 1)) * B * H + (long long)(cu * 4 + wave) * 512 + lane * 8;
            if (a.mode & 32)
                asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst), "v"(v) : "memory");
            else
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // EVERY storing wave drains
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        STAMP(4);
        // ---- off the critical path: Y_t and c_t rows (plain 16-byte stores), next step's pre-activations
        {
            const int row = threadIdx.x >> 2, ch = threadIdx.x & 3;
            const uint4 y = *reinterpret_cast<const uint4*>(hstage + (row * UPC + ch * 8) * 2);
            *reinterpret_cast<uint4*>(a.Y + (((long long)inst * a.T + t) * B + row) * H + cu * UPC + ch * 8) = y;
            const uint4 c0 = *reinterpret_cast<const uint4*>(cstage + row * UPC + ch * 8);
            const uint4 c1 = *reinterpret_cast<const uint4*>(cstage + row * UPC + ch * 8 + 4);
            float* cd = a.C + (((long long)inst * a.T + t) * B + row) * H + cu * UPC + ch * 8;
            *reinterpret_cast<uint4*>(cd) = c0;
            *reinterpret_cast<uint4*>(cd + 4) = c1;
            if (t + 1 < a.T) {
                gq0 = ld_g(a.G, a.T, inst, t + 1, cu, wave, lane, 0);
                gq1 = ld_g(a.G, a.T, inst, t + 1, cu, wave, lane, 1);
                gq2 = ld_g(a.G, a.T, inst, t + 1, cu, wave, lane, 2);
                gq3 = ld_g(a.G, a.T, inst, t + 1, cu, wave, lane, 3);
            }
        }
        STAMP(5);
    }
    if (inst == 0 && cu == 0 && threadIdx.x == 0)
        for (int i = 0; i < 7; ++i) a.phase[i] = ph[i];
    if (cu == 0 && threadIdx.x == 0) a.cycles[inst] = wall_clock64() - t_begin;
}

// ---------------------------------------------------------------- host
static float bf2f_h(bf16_t b) {
    unsigned u = ((unsigned)b) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static bf16_t f2bf_h(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
static unsigned rng_state = 12345u;
static float urand() {
    rng_state = rng_state * 1664525u + 1013904223u;
    return (rng_state >> 8) * (1.0f / 16777216.0f);
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 400;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    const int Tcheck = 6;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, wall clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);

    // one weight matrix shared by all instances (each instance gets its own packed copy)
    std::vector<bf16_t> W((size_t)4 * H * H);
    for (auto& x : W) x = f2bf_h((urand() * 2 - 1) / 32.f);
    std::vector<bf16_t> Wreg((size_t)NINST * CUS * 4 * 2 * KS * 64 * 8);
    for (int inst = 0; inst < NINST; ++inst)
        for (int cu = 0; cu < CUS; ++cu)
            for (int wv = 0; wv < 4; ++wv)
                for (int tn = 0; tn < 2; ++tn)
                    for (int ks = 0; ks < KS; ++ks)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 8; ++e) {
                                const int nn = l & 15, gate = tn * 2 + (nn >> 3);
                                const int unit = cu * UPC + wv * 8 + (nn & 7);
                                const int k = ks * 32 + (l >> 4) * 8 + e;
                                const size_t o = ((((((size_t)inst * CUS + cu) * 4 + wv) * 2 + tn) * KS + ks) * 64 + l) * 8 + e;
                                Wreg[o] = W[((size_t)gate * H + unit) * H + k];
                            }
    // pre-activations in the lane order the kernel reads: [inst][T][cu][wave][lane][cell 8][gate 4]
    // plain copy Gp[t][row][gate][unit] for the CPU check (same values for every instance)
    std::vector<bf16_t> Gp((size_t)T * B * 4 * H);
    for (auto& x : Gp) x = f2bf_h((urand() * 2 - 1) * 1.5f);
    std::vector<bf16_t> G((size_t)NINST * T * CUS * 4 * 64 * 32);
    for (int inst = 0; inst < NINST; ++inst)
        for (int t = 0; t < T; ++t)
            for (int cu = 0; cu < CUS; ++cu)
                for (int wv = 0; wv < 4; ++wv)
                    for (int l = 0; l < 64; ++l) {
                        const int nn = l & 15, qq = l >> 4;
                        const bool lo = nn < 8;
                        const int unit = cu * UPC + wv * 8 + (nn & 7);
                        for (int rb = 0; rb < 4; ++rb)
                            for (int r = 0; r < 2; ++r) {
                                const int row = rb * 16 + qq * 4 + (lo ? 0 : 2) + r, cell = rb * 2 + r;
                                for (int g = 0; g < 4; ++g)
                                    G[((((((size_t)inst * T + t) * CUS + cu) * 4 + wv) * 64 + l) * 8 + cell) * 4 + g] =
                                        Gp[(((size_t)t * B + row) * 4 + g) * H + unit];
                            }
                    }

    Args a;
    bf16_t *dW, *dG, *dh, *dY;
    float* dC;
    unsigned *dcnt, *dtick, *derr;
    long long* dcyc;
    long long* dph;
    CK(hipMalloc(&dW, Wreg.size() * 2));
    CK(hipMalloc(&dG, G.size() * 2));
    CK(hipMalloc(&dh, (size_t)NINST * 2 * B * H * 2));
    CK(hipMalloc(&dY, (size_t)NINST * T * B * H * 2));
    CK(hipMalloc(&dC, (size_t)NINST * T * B * H * 4));
    CK(hipMalloc(&dcnt, 64));
    CK(hipMalloc(&dtick, 64));
    CK(hipMalloc(&derr, 64));
    CK(hipMalloc(&dcyc, 64));
    CK(hipMalloc(&dph, 64));
    CK(hipMemset(dph, 0, 64));
    CK(hipMemcpy(dW, Wreg.data(), Wreg.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dG, G.data(), G.size() * 2, hipMemcpyHostToDevice));
    a.Wreg = dW; a.G = dG; a.himg = dh; a.Y = dY; a.C = dC; a.counter = dcnt; a.ticket = dtick;
    a.err = derr; a.cycles = dcyc; a.phase = dph; a.T = T; a.mode = mode;
    const size_t lds_bytes = IMG_BYTES + 4096 + 8192;
    CK(hipFuncSetAttribute((const void*)wsr_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(dh, 0, (size_t)NINST * 2 * B * H * 2));
        CK(hipMemset(dcnt, 0, 64));
        CK(hipMemset(dtick, 0, 64));
        CK(hipMemset(derr, 0, 64));
        CK(hipMemset(dcyc, 0, 64));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(wsr_probe, dim3(256), dim3(256), lds_bytes, 0, a);
        CK(hipGetLastError());
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned err[16];
        long long cyc[8];
        CK(hipMemcpy(err, derr, 64, hipMemcpyDeviceToHost));
        CK(hipMemcpy(cyc, dcyc, 64, hipMemcpyDeviceToHost));
        printf("rep %d mode %d: T=%d  %.3f ms total  %.2f us/step  err=%u  census", rep, mode, T, ms, 1e3 * ms / T, err[0]);
        for (int x = 0; x < 8; ++x) printf(" %u", err[1 + x]);
        printf("  | wall-clock ticks/step inst0 %.0f\n", (double)cyc[0] / T);
        long long ph[8];
        CK(hipMemcpy(ph, dph, 64, hipMemcpyDeviceToHost));
        printf("      phases (10 ns ticks/step, inst 0 cu 0): wait %.0f gather %.0f mfma %.0f cell %.0f publish %.0f stores+prefetch %.0f\n",
               (double)ph[0] / T, (double)ph[1] / T, (double)ph[2] / T, (double)ph[3] / T, (double)ph[4] / T, (double)ph[5] / T);
    }

    // ---- correctness (mode 0 only): first Tcheck steps of instance 0 and 5 against a CPU recurrence
    if (!(mode & 3)) {
        std::vector<float> h((size_t)B * H, 0.f), cst((size_t)B * H, 0.f), hn((size_t)B * H);
        std::vector<float> Wf(W.size());
        for (size_t i = 0; i < W.size(); ++i) Wf[i] = bf2f_h(W[i]);
        std::vector<bf16_t> Yd((size_t)Tcheck * B * H);
        std::vector<float> Cd((size_t)Tcheck * B * H);
        double worst = 0;
        for (int which = 0; which < ((mode & 4) ? 1 : 2); ++which) {
            const int inst = which == 0 ? 0 : 5;
            CK(hipMemcpy(Yd.data(), dY + (size_t)inst * T * B * H, Yd.size() * 2, hipMemcpyDeviceToHost));
            CK(hipMemcpy(Cd.data(), dC + (size_t)inst * T * B * H, Cd.size() * 4, hipMemcpyDeviceToHost));
            std::fill(h.begin(), h.end(), 0.f);
            std::fill(cst.begin(), cst.end(), 0.f);
            for (int t = 0; t < Tcheck; ++t) {
                for (int b = 0; b < B; ++b)
                    for (int u = 0; u < H; ++u) {
                        float pre[4];
                        for (int g = 0; g < 4; ++g) {
                            const float* wr = &Wf[((size_t)g * H + u) * H];
                            const float* hr = &h[(size_t)b * H];
                            float s = 0.f;
                            for (int k = 0; k < H; ++k) s += wr[k] * hr[k];
                            pre[g] = s + bf2f_h(Gp[(((size_t)t * B + b) * 4 + g) * H + u]);
                        }
                        const float gi = 1.f / (1.f + expf(-pre[0])), gf = 1.f / (1.f + expf(-pre[1]));
                        const float gg = tanhf(pre[2]), go = 1.f / (1.f + expf(-pre[3]));
                        const float cn = gf * cst[(size_t)b * H + u] + gi * gg;
                        cst[(size_t)b * H + u] = cn;
                        hn[(size_t)b * H + u] = bf2f_h(f2bf_h(go * tanhf(cn)));
                    }
                h = hn;
                double wy = 0, wc = 0;
                for (size_t i = 0; i < (size_t)B * H; ++i) {
                    wy = fmax(wy, fabs(bf2f_h(Yd[(size_t)t * B * H + i]) - h[i]));
                    wc = fmax(wc, fabs(Cd[(size_t)t * B * H + i] - cst[i]));
                }
                printf("  inst %d step %d: max |dY| %.4g  max |dC| %.4g\n", inst, t, wy, wc);
                worst = fmax(worst, fmax(wy, wc));
            }
        }
        printf("check vs CPU recurrence (%d steps): max |diff| = %.4g  %s\n", Tcheck, worst, worst < 3e-2 ? "OK" : "MISMATCH");
    }
    return 0;
}
