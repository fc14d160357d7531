// bf16 "TN" GEMM for weight gradients, 256 x 128 macro-tiles, fp32 K-slice partials:
//
//     P[s][M,N] = sum over the K rows of slice s of  A[k, m] * B[k, n]        (A: [K, lda], B: [K, ldb])
//
// i.e. dW = dY^T X with both operands in their natural activation layout (rows = frames, the reduction
// runs over rows): the encoder stack's dW_ih / dW_hh (rnnt/models.py:56-66 through autograd) and the
// joint's dW2 (rnnt/models.py:165-167).  These products run in the BACKGROUND beside the launch-bound BPTT
// recurrence, which is bound by the same CU -> L2 fetch path, so what matters is bytes pulled per flop and
// leaving room on the CU, not peak rate:
//
//   * 256 x 128 tiles (85 flop per operand byte; the register-staged 128 x 128 kernel of gemm.hip: 64),
//     8 waves of 64 x 64 (<= 128 registers each), ONE workgroup per CU, 96 KB of LDS: half of the CU's
//     registers and 64 KB of LDS stay free, so a recurrence workgroup still fits beside it (an 8-wave
//     256 x 256 tile would own the CU).  Eight waves, not four with bigger wave tiles: the LDS-DMA path
//     moves ~21 GB/s per CU with four waves issuing and ~2x that with eight, whatever the ring depth
//     (measured: the 4-wave version of this kernel stayed at 290 TFLOP/s on the joint's dW2 with two 64-k
//     stages and with four 32-k stages; waves parked 55 % of the time, no bank conflicts);
//   * operands go HBM -> LDS directly (global_load_lds_dwordx4) in their row-major [k][m] form - the DMA
//     cannot transpose - as panels of [32 k][64 m] (128-byte rows, 16-byte chunk index XOR-ed with k & 7 on
//     the SOURCE side), in a ring of four 32-k stages (24 KB each) with COUNTED waits: up to three stages
//     (72 KB) are in flight per CU;
//   * the MFMA fragments (8 k per lane for one m) come out of LDS through the gfx950 transpose read
//     `ds_read_b64_tr_b16`: a 16-lane group reads a [4 k][16 m] block (lane i supplies the address of 4
//     consecutive m of row k = i / 4) and lane c receives the 4 k values of column c.  Two reads make a
//     fragment; the k ORDER inside a 32-k MFMA step is therefore permuted (lane group g holds k = 4 g + j
//     and 16 + 4 g + j) - identically for both operands, which is all a dot product needs;
//   * QUIET: a workgroup walks its (slice, tile) items and writes each fp32 partial tile exactly once with
//     plain stores - no atomics, no dirty lines while the main loop runs (boundary_probe: a concurrent
//     writer makes every dependent launch boundary of the recurrence 3-10 x dearer); the caller's reduce
//     pass sums the slices (gemm.hip reduce_partials_kernel / encoder_stack.hip unpermute_rows_kernel).
//
// Requirements (else gemm.hip runs): M % 8 == 0, N % 8 == 0, lda % 8 == 0, ldb % 8 == 0, 16-byte aligned
// operands.  K is arbitrary: rows past the end of a slice are read from a zero line.
#include <stdlib.h>

#include "common.hpp"
#include "gemm_nt.hpp"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int TM = 256, TN = 128, BK = 32;
constexpr int A_ST = BK * TM * 2;             // A stage: 32 k x 256 m = 16 KB (4 panels of [32 k][64 m], 4 KB each)
constexpr int B_ST = BK * TN * 2;             // B stage: 8 KB (2 panels)
constexpr int STAGE = A_ST + B_ST;            // 24 KB
constexpr int NSTAGE = 4;                     // ring: one stage being read, up to three in flight
constexpr int LDS_BYTES = NSTAGE * STAGE;     // 96 KB

__device__ uint4 g_zero_line[4];              // 64 zero bytes: the source of K rows past a slice's end

struct Tn256Args {
    const bf16_t* A;
    const bf16_t* B;
    float* P;
    long long lda, ldb, pstride;
    int M, N, K;
    int n_tiles, tiles, slices, k_per, items, per;
};

__device__ __forceinline__ void glds16(const bf16_t* src, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ bf16x8_t tr_frag(const unsigned char* p0, const unsigned char* p1) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p1);
    const s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
}

__global__ __launch_bounds__(512, 2) void gemm_tn256_kernel(Tn256Args g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;        // 8 waves: wave rows wm*64, wave cols wn*64
    const int r16 = lane & 15, kq = lane >> 4;

    // ---- DMA piece of this lane: k row wave*8 + (lane >> 3) of a 32-k stage, LDS chunk lane & 7 of
    // the 128-byte panel row, which holds SOURCE chunk (lane & 7) ^ (k & 7)
    const int krow = (wave & 3) * 8 + (lane >> 3);
    const int pw = wave >> 2;                       // this wave brings A panels pw and 2 + pw, B panel pw
    const int schunk = (lane & 7) ^ (lane >> 3);

    // ---- transpose-read addresses inside a stage's panel (see the header): read h of a fragment, lane
    // (g = kq, i = r16) points at row k = 16 h + 4 g + (i >> 2), 4 consecutive m from (i & 3) * 4 of the
    // fragment's 16; fragment f of a panel (f & 3) XORs the chunk index with 2 (f & 3)
    int tr_off[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k = 16 * h + 4 * kq + (r16 >> 2);
        tr_off[h] = k * 128 + (((((r16 & 3) >> 1) ^ (k & 7))) << 4) + (r16 & 1) * 8;
    }

    const int G = gridDim.x, b = blockIdx.x;
    const int pos = (G % 8 == 0) ? (b % 8) * (G / 8) + b / 8 : b;   // one XCD walks adjacent items
    const int item_end = min(g.items, (pos + 1) * g.per);
    for (int item = pos * g.per; item < item_end; ++item) {
        const int s = item / g.tiles, t = item % g.tiles;
        const int m0 = (t / g.n_tiles) * TM, n0 = (t % g.n_tiles) * TN;
        const int kbeg = s * g.k_per, kend = min(g.K, kbeg + g.k_per);
        const int NQ = (kend - kbeg + BK - 1) / BK;

        const bf16_t* asrc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) asrc[i] = g.A + min(m0 + (2 * i + pw) * 64 + schunk * 8, g.M - 8);
        const bf16_t* bsrc = g.B + min(n0 + pw * 64 + schunk * 8, g.N - 8);
        const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_line);

        // stage q (32 k rows): 24 pieces of 1 KB (16 of A, 8 of B), 3 DMA instructions per lane
        auto issue = [&](int q) {
            unsigned char* base = smem + (q & (NSTAGE - 1)) * STAGE + (wave & 3) * 1024;
            const int k = kbeg + q * BK + krow;
            const bool in = k < kend;
            const long long ra = (long long)k * g.lda, rb = (long long)k * g.ldb;
#pragma unroll
            for (int i = 0; i < 2; ++i) glds16(in ? asrc[i] + ra : zero, base + (2 * i + pw) * 4096);
            glds16(in ? bsrc + rb : zero, base + A_ST + pw * 4096);
        };

        f32x4_t acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

        // ---- main loop.  Stage q is one phase [LOAD block | barrier | 16 MFMA | barrier]; the two waves that
        // share a SIMD (wave w and w + 4) run ONE barrier apart, so while one group issues its LDS reads and
        // DMA instructions the other owns the matrix pipe (lock-step waves would do LOAD, then LDS, then MFMA
        // all at the same time: 311 TFLOP/s on the joint's dW2).  LOAD(q): transpose-read the fragments of
        // stage q, issue stage q + 3 into the slot of stage q - 1 (last read one phase ago by this group, and
        // before the barrier this block started behind by the other), then a COUNTED wait that leaves the two
        // youngest stages in flight: stage q + 1 has landed, one barrier before this group and two before
        // the other group read it.
        const int grp = wave >> 2;
        __syncthreads();             // the previous item's LDS reads are done
        issue(0);
        if (NQ > 1) issue(1);
        if (NQ > 2) issue(2);
        if (NQ > 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (NQ > 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (grp == 1) asm volatile("s_barrier" ::: "memory");
        for (int q = 0; q < NQ; ++q) {
            const unsigned char* sA = smem + (q & (NSTAGE - 1)) * STAGE + wm * 4096;
            const unsigned char* sB = smem + (q & (NSTAGE - 1)) * STAGE + A_ST + wn * 4096;
            bf16x8_t bf[4], af[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bf[j] = tr_frag(sB + (tr_off[0] ^ (j << 5)), sB + (tr_off[1] ^ (j << 5)));
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[i] = tr_frag(sA + (tr_off[0] ^ (i << 5)), sA + (tr_off[1] ^ (i << 5)));
            if (q + 3 < NQ) issue(q + 3);
            const int younger = min(NQ - 1, q + 3) - (q + 1);     // stages issued after stage q + 1
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            asm volatile("s_barrier" ::: "memory");
        }
        if (grp == 0) asm volatile("s_barrier" ::: "memory");
        // ---- the partial tile leaves once, plain 16-byte stores (lane: row m, 4 consecutive n)
        float* P = g.P + (long long)s * g.pstride;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m0 + wm * 64 + i * 16 + r16;
            if (row >= g.M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = n0 + wn * 64 + j * 16 + kq * 4;
                if (col >= g.N) continue;
                *reinterpret_cast<float4*>(P + (long long)row * g.N + col) =
                    make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        }
    }
}

}  // namespace

bool ed_gemm_tn256_ok(const void* A, long long lda, const void* B, long long ldb, int M, int N, int K) {
    static const int on = [] { const char* e = getenv("EDGEDICT_GEMM_TN256"); return e ? atoi(e) : 1; }();
    return on && A && B && M >= 8 && N >= 8 && K >= 1 && M % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 &&
           (uintptr_t)A % 16 == 0 && (uintptr_t)B % 16 == 0;
}

// K slices that fill the chip with one workgroup per CU (at most max_slices, at least 8 K stages each)
int ed_gemm_tn256_slices(int M, int N, int K, int max_slices) {
    const long long tiles = (long long)((M + TM - 1) / TM) * ((N + TN - 1) / TN);
    int n_cu = 256, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
        n_cu = 256;
    long long s = n_cu / (tiles > 0 ? tiles : 1);
    if (s > max_slices) s = max_slices;
    const int stages = (K + BK - 1) / BK;
    while (s > 1 && s * 8 > stages) --s;
    return s < 1 ? 1 : (int)s;
}

int ed_gemm_tn256_partials(const void* A, long long lda, const void* B, long long ldb, float* partials, int M,
                           int N, int K, int slices, int max_wgs, hipStream_t s) {
    ED_CHECK_ARG(partials && slices >= 1 && slices <= 64, "gemm_tn256: bad partials / slices");
    Tn256Args g;
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.P = partials;
    g.lda = lda; g.ldb = ldb; g.pstride = (long long)M * N;
    g.M = M; g.N = N; g.K = K;
    g.n_tiles = (N + TN - 1) / TN;
    const long long tiles = (long long)((M + TM - 1) / TM) * g.n_tiles;
    ED_CHECK_ARG(tiles * slices < (1ll << 30), "gemm_tn256: too many tiles");
    g.tiles = (int)tiles;
    g.slices = slices;
    g.k_per = ((K + slices - 1) / slices + 63) / 64 * 64;
    g.items = g.tiles * slices;
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            n = 256;
        return n > 0 ? n : 256;
    }();
    int grid = n_cu;                     // one workgroup per CU is resident (96 KB LDS)
    if (max_wgs > 0 && max_wgs < grid) grid = max_wgs;
    if (grid > g.items) grid = g.items;
    if (grid > 8) grid = grid / 8 * 8;
    g.per = (g.items + grid - 1) / grid;
    grid = (g.items + g.per - 1) / g.per;
    ED_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_tn256_kernel,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipLaunchKernelGGL(gemm_tn256_kernel, dim3((unsigned)grid), dim3(512), LDS_BYTES, s, g);
    ED_CHECK_LAUNCH("gemm_tn256");
    return ED_OK;
}
