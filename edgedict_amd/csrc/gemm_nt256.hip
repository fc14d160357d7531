// bf16 "NT" GEMM, 256 x 256 macro-tiles:  C[M,N] = A[M,K] * B[N,K]^T + bias1[N] + bias2[N]  (bf16 out).
//
// The large products of the path - joint logits hid x W2^T (rnnt/models.py:165-167,177), its
// input gradient through a W2^T copy, whole-layer input products - are MFMA-bound, and the
// 128 x 128 kernel of gemm_nt.hip spends half of a short-K tile in prologue/epilogue and fetches
// 64 flop per operand byte.  This kernel doubles both tile edges (128 flop/B) and never drains its
// DMA pipeline inside the K loop:
//
//   * 8 waves (2 along M x 4 along N), wave tile 128 x 64 = 8 x 4 MFMA 16x16x32 tiles (128 accumulator
//     registers), ONE workgroup per CU (128 KB LDS);
//   * operands go HBM -> LDS directly (global_load_lds_dwordx4) in HALF-TILES of 256 rows x 32 k
//     (16 KB, two DMA instructions per lane).  A K tile (BK = 64) is four half-tiles:
//     A.k0, B.k0, A.k1, B.k1; while the MFMAs of k-half h of K tile t run, the four half-tiles of K
//     tile t+1 are issued one per MFMA block, and the only waits are COUNTED (`s_waitcnt vmcnt(4)`:
//     the two newest half-tiles stay in flight) followed by a raw s_barrier - a plain
//     __syncthreads() would emit vmcnt(0) and drain the DMA queue (cdna_hip_programming.md 5);
//   * 64-byte LDS rows would put the 16 rows of a fragment read on 4 banks groups 4 times: the 16-byte
//     chunk index is XOR-ed with (row >> 2) & 3 - on the SOURCE address, the DMA image is lane-linear;
//   * operand-swapped MFMA (as gemm_nt.hip): a lane holds 4 consecutive columns of one C row, the
//     C tile is staged through the (dead) operand buffers and leaves in 16-byte row segments;
//     biases enter through the accumulators.
//
// Requirements (else gemm_nt.hip / gemm.hip run): K % 64 == 0, K >= 128, lda/ldb/ldc % 8 == 0,
// N % 8 == 0, 16-byte aligned operands, no accumulate.
#include <stdlib.h>

#include "common.hpp"
#include "gemm_nt.hpp"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int TM = 256, TN = 256, BK = 64;
constexpr int HT = 256 * 32 * 2;          // one half-tile: 256 rows x 32 k bf16 = 16 KB
constexpr int BUF = 4 * HT;               // A.k0 | A.k1 | B.k0 | B.k1
constexpr int LDS_BYTES = 2 * BUF;        // 128 KB

struct Nt256Args {
    const bf16_t* A;
    const bf16_t* B;
    bf16_t* C;
    const float* bias1;
    const float* bias2;
    long long lda, ldb, ldc;
    int M, N, K;
    int n_tiles, tiles;
    int dbg;            // EDGEDICT_NT256_DEBUG ablation bits (tools/nt256_ablate.py): 1 no lse, 2 no C store, 4 no MFMA, 8 no C staging
    float2* lse_part;   // optional [M][lse_slots] (max, sum exp(x - max)) over 64-column slots of each C row
    int lse_slots;
};

__device__ __forceinline__ void glds16(const bf16_t* src, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// One tile per workgroup, C staged through the (dead) operand buffers and stored in 16-byte row segments.  (A
// persistent tile loop - 256 workgroups walk the tiles, the finished tile leaves straight from the accumulators, the
// next tile's DMA prologue is issued at once - was built and measured in round 2: logits 2.19 vs 2.14 ms, dhid 1.15
// vs 1.15, step 25.95 vs 25.87 ms, also with the CUs' tile phases spread at the start: the epilogue's cost is issue
// and VALU work that only a second accumulator set could overlap, not store latency.  Removed in round 4.)
__global__ __launch_bounds__(512, 1) void gemm_nt256_kernel(Nt256Args g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;        // wave rows wm*128, wave cols wn*64
    const int r16 = lane & 15, kq = lane >> 4;
    const int KT = g.K / BK;

    // XCD-aware tile order (bijective): the workgroups of one XCD (blockIdx % 8) walk adjacent tiles of that
    // XCD's contiguous tile range, so the A row panel they share stays in its L2
    const int nx = 8, xq = g.tiles / nx, xr = g.tiles % nx, xcd = blockIdx.x % nx, iw = blockIdx.x / nx;
    const int x_start = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
    const int x_count = xq + (xcd < xr ? 1 : 0);
    if (iw >= x_count) return;
  {
    const int tile = x_start + iw;
    const int m0 = (tile / g.n_tiles) * TM, n0 = (tile % g.n_tiles) * TN;

    // ---- DMA sources.  A half-tile is 16 pieces of 16 rows x 64 bytes; wave w brings pieces w and w+8.
    // Lane l of a piece: row l >> 2, LDS chunk l & 3, which holds SOURCE chunk (l & 3) ^ ((row >> 2) & 3).
    const int prow = lane >> 2;
    const int schunk = (lane & 3) ^ ((prow >> 2) & 3);
    const bf16_t* asrc[2];
    const bf16_t* bsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave + 8 * i) * 16 + prow;
        asrc[i] = g.A + (long long)min(m0 + row, g.M - 1) * g.lda + schunk * 8;
        bsrc[i] = g.B + (long long)min(n0 + row, g.N - 1) * g.ldb + schunk * 8;
    }
    // half-tile h of K tile kt: h = 0 A.k0, 1 B.k0, 2 A.k1, 3 B.k1 (issue order)
    auto issue = [&](int kt, int h) {
        unsigned char* base = smem + (kt & 1) * BUF + ((h & 1) * 2 + (h >> 1)) * HT;
        const int k0 = kt * BK + (h >> 1) * 32;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            glds16(((h & 1) ? bsrc[i] : asrc[i]) + k0, base + (wave + 8 * i) * 1024);
    };

    // ---- the accumulators start at the biases (ordinary loads, before any DMA is in flight)
    f32x4_t acc[8][4];
    float4 bv[4], bw[4];     // kept apart until they are used: adding them here would wait for the loads
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nc = n0 + wn * 64 + j * 16 + kq * 4;
        bv[j] = bw[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (nc < g.N) {
            if (g.bias1) bv[j] = *reinterpret_cast<const float4*>(g.bias1 + nc);
            if (g.bias2) bw[j] = *reinterpret_cast<const float4*>(g.bias2 + nc);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[i][j] = (f32x4_t){bv[j].x + bw[j].x, bv[j].y + bw[j].y, bv[j].z + bw[j].z, bv[j].w + bw[j].w};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const int a_off = (wm * 128 + r16) * 64 + ((kq ^ ((r16 >> 2) & 3)) << 4);   // + i * 1024 per m-fragment
    const int b_off = (wn * 64 + r16) * 64 + ((kq ^ ((r16 >> 2) & 3)) << 4);    // + j * 1024 per n-fragment

    // ---- main loop: 4 phases per K tile, phase (kh, mh) = [LOAD block | barrier | 16 MFMA | barrier].
    // The two waves that share a SIMD (wave w and w + 4, i.e. wm = 0 / 1) run ONE barrier apart, so
    // between two barriers one of them is in its LOAD block (LDS reads, DMA issue) while the other owns
    // the matrix pipe.  Staging: phase p = 2 kh + mh of K tile kt issues half-tile p of K tile kt + 1
    // (order A.k0, B.k0, A.k1, B.k1).  The counted waits sit in the LOAD blocks of phases 1 and 3, two
    // barriers ahead of the first read of what they retire (one barrier more than lock-step waves
    // would need: the partner group waits one barrier later).  A slot is re-staged >= 3 phases after
    // its last read.
#pragma unroll
    for (int h = 0; h < 4; ++h) issue(0, h);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (wm == 1) asm volatile("s_barrier" ::: "memory");

    bf16x8_t b[4], a[4];
    for (int kt = 0; kt < KT; ++kt) {
        const unsigned char* buf = smem + (kt & 1) * BUF;
        const bool more = kt + 1 < KT;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const unsigned char* sA = buf + kh * HT;
            const unsigned char* sB = buf + (2 + kh) * HT;
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
                // ---- LOAD block
                if (mh == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(sB + b_off + j * 1024);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    a[i] = *reinterpret_cast<const bf16x8_t*>(sA + a_off + (mh * 4 + i) * 1024);
                if (more) issue(kt + 1, 2 * kh + mh);
                if (mh == 1) {
                    if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                asm volatile("s_barrier" ::: "memory");
                // ---- MFMA block
                __builtin_amdgcn_s_setprio(1);
                if (!(g.dbg & 4))
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[mh * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[mh * 4 + i][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                asm volatile("s_barrier" ::: "memory");
            }
        }
    }
    if (wm == 0) asm volatile("s_barrier" ::: "memory");

    constexpr int CCH = TN / 8;      // 16-byte chunks per staged C row
    unsigned char* sC = smem;        // [256 rows][32 chunks], chunk ^= row & 31
    // ---- epilogue: the operand buffers are dead after this barrier; C is staged in LDS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (!(g.dbg & 8))
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nl = wn * 64 + j * 16 + kq * 4;          // 4 consecutive columns nl .. nl+3
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ml = wm * 128 + i * 16 + r16;
            uint2 pk;
            pk.x = f32x2_to_bf16x2(acc[i][j][0], acc[i][j][1]);
            pk.y = f32x2_to_bf16x2(acc[i][j][2], acc[i][j][3]);
            *reinterpret_cast<uint2*>(sC + ml * (TN * 2) + ((((nl >> 3) ^ (ml & (CCH - 1))) << 4) | ((nl & 4) << 1))) = pk;
        }
    }
    // ---- optional fused log-sum-exp partials (the RNN-T loss needs log_softmax denominators of every
    // logits row, rnnt/models.py:238 -> warprnnt): per C row, (max, sum exp(x - max)) over this wave's 64
    // columns, computed from the bf16-ROUNDED values (exactly what a later pass over the stored logits
    // would see), reduced over the 4 lanes that share a row, one 8-byte store per (row, slot).  The
    // denominators are then finished from M x N/64 pairs instead of a second pass over M x N logits.
    if (g.lse_part && !(g.dbg & 1)) {
        constexpr float LOG2E = 1.4426950408889634f;
        const bool ragged = n0 + TN > g.N;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v[16];
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int nc = n0 + wn * 64 + j * 16 + kq * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float x = bf16_to_f32(f32_to_bf16(acc[i][j][q]));
                    if (ragged && nc + q >= g.N) x = -INFINITY;
                    v[j * 4 + q] = x;
                    mx = fmaxf(mx, x);
                }
            }
            // sum exp(x - mx) = sum exp2(x * log2e - mx * log2e): one fma + one v_exp_f32 per value; a row of
            // -inf only (ragged edge) sums to 0
            const float mb = (mx == -INFINITY) ? 0.f : mx * LOG2E;
            float sm = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) sm += __builtin_amdgcn_exp2f(fmaf(v[e], LOG2E, -mb));
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
                const float om = __shfl_xor(mx, off, 64), os = __shfl_xor(sm, off, 64);
                const float nm = fmaxf(mx, om);
                sm = (nm == -INFINITY) ? 0.f : sm * __expf(mx - nm) + os * __expf(om - nm);
                mx = nm;
            }
            const int row = m0 + wm * 128 + i * 16 + r16;
            const int slot = (n0 >> 6) + wn;
            if (kq == 0 && row < g.M && slot < g.lse_slots)
                g.lse_part[(long long)row * g.lse_slots + slot] = make_float2(mx, sm);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (!(g.dbg & 2))
#pragma unroll 4
    for (int it = 0; it < TM * TN / 8 / 512; ++it) {
        const int c = threadIdx.x + it * 512;
        const int rl = c / CCH, ch = c % CCH;
        const int row = m0 + rl, col = n0 + ch * 8;
        if (row >= g.M || col >= g.N) continue;
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(sC + rl * (TN * 2) + ((ch ^ (rl & (CCH - 1))) << 4));
        bf16_t* dst = g.C + (long long)row * g.ldc + col;
        // C is written once and not read by this launch: keep it OUT of the L2 that holds the A row panels and B
        // (write-through sc1, the line is dropped: logits 1.99 -> 1.92 ms; non-temporal stores the same; bit 16
        // of EDGEDICT_NT256_DEBUG restores plain stores)
        if (g.dbg & 16) *reinterpret_cast<u32x4_t*>(dst) = v;
        else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
    }
  }
}

}  // namespace

bool ed_gemm_nt256_shape_ok(int M, int N, int K) {
    return M > 0 && N > 0 && K >= 128 && K % 64 == 0;
}

bool ed_gemm_nt256_ok(int M, int N, int K, int accumulate) {
    static const int on = [] { const char* e = getenv("EDGEDICT_GEMM_NT256"); return e ? atoi(e) : 1; }();
    // worth it from ~2 tiles per CU on; the 128 x 128 kernel keeps the small and the accumulating products
    return on && !accumulate && ed_gemm_nt256_shape_ok(M, N, K) &&
           (long long)((M + 255) / 256) * ((N + 255) / 256) >= 512;
}

int ed_gemm_nt256_launch(const void* A, long long lda, const void* B, long long ldb, void* C,
                         long long ldc, int M, int N, int K, const float* bias1, const float* bias2,
                         hipStream_t s, float* lse_part) {
    if (ed_gemm_nt256r_ok(M, N, K, bias1 || bias2))
        return ed_gemm_nt256r_launch(A, lda, B, ldb, C, ldc, M, N, K, bias1, bias2, s, lse_part);
    Nt256Args g;
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = (bf16_t*)C;
    g.bias1 = bias1; g.bias2 = bias2;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K;
    g.lse_part = (float2*)lse_part;
    static const int dbg = [] { const char* e = getenv("EDGEDICT_NT256_DEBUG"); return e ? atoi(e) : 0; }();
    g.dbg = dbg;
    g.lse_slots = (N + 63) / 64;
    g.n_tiles = (N + TN - 1) / TN;
    const long long tiles = (long long)((M + TM - 1) / TM) * g.n_tiles;
    ED_CHECK_ARG(tiles < (1ll << 31), "gemm: too many tiles");
    g.tiles = (int)tiles;
    ED_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_nt256_kernel,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipLaunchKernelGGL(gemm_nt256_kernel, dim3((unsigned)tiles), dim3(512), LDS_BYTES, s, g);
    ED_CHECK_LAUNCH("gemm_nt256");
    return ED_OK;
}
