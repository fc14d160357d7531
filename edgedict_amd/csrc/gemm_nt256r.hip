// bf16 "NT" GEMM, 256 x 256 macro-tiles, PERSISTENT with an 8-slot operand ring (round 6):
//   C[M,N] = A[M,K] * B[N,K]^T + bias1[N] + bias2[N]   (bf16 out; optional fused log-sum-exp partials)
//
// Same products as gemm_nt256.hip (joint logits hid x W2^T, rnnt/models.py:165-167,177; its input gradient
// through a W2^T copy; whole-layer input products) and bit-identical results (same MFMA order over K).  What
// round 5's profile showed about that kernel: half of a short-K tile is prologue + epilogue with the matrix
// pipe idle (one workgroup per CU, 128 KB LDS), and its K loop runs at the rate its operand fetch allows -
// 64-byte row pieces (every 128-byte line requested twice) with 32 KB in flight per CU.  This kernel changes
// the three things that follow from it:
//
//   * operand pieces are 8 rows x 128 BYTES (full lines; source-side XOR swizzle chunk ^= row & 7, the
//     fragment reads stay conflict-free); a K tile (BK = 64) is four 16 KB half-tiles A0 A1 (rows 0-127 /
//     128-255 of the A tile) B0 B1 (for each n-wave the first / second 32 of its 64 columns);
//   * the 128 KB of LDS are a RING of 8 half-tile slots ([K-tile parity][A0 A1 B0 B1]).  A wave tile is
//     walked in four quadrant phases (m-half, n-half) = (0,0) (0,1) (1,1) (1,0), 16 MFMA each (4 x 2
//     fragments x 2 k-steps); every slot is re-staged two phases after its last read with the half-tile of
//     the K tile TWO ahead: one half-tile issue per phase, 4-5 half-tiles (64-80 KB) in flight, counted
//     `s_waitcnt vmcnt(8)` + raw s_barrier, two wave groups one barrier apart as before;
//   * the workgroup is persistent (<= 256 workgroups walk the tiles; XCD-contiguous ranges, the 32
//     workgroups of an XCD on 32 adjacent tiles at any time) and the ring does not know about tile
//     boundaries: while a tile's epilogue runs, the first two K tiles of the next tile are already in
//     flight (no prologue bubble).  C leaves straight from the accumulators: lane pairs swap two packed
//     fragments with v_permlane16_swap so that each lane owns a 16-byte chunk; a wave's two stores of a row
//     fragment complete 128-byte lines.  LDS is never used by the epilogue.
//
// vmcnt discipline: VMEM loads, LDS-DMA and stores retire in issue order on gfx9-family hardware (one
// counter, which is why the compiler itself emits vmcnt(N > 0) after load + store sequences), so the
// epilogue's stores only shift the immediates of the first K tile that follows them: the K-tile body
// exists in two copies, <0> (nothing but DMA in flight) and <EPI_OPS> (a FULL tile's epilogue - a fixed
// number of unconditional stores - was issued just before).  Ragged tiles use predicated stores and are
// followed by the strict copy.
//
// Requirements (else gemm_nt256.hip / gemm_nt.hip / gemm.hip run): K % 64 == 0, K >= 128, lda/ldb/ldc % 8
// == 0, N % 8 == 0, 16-byte aligned operands, no accumulate; with a bias, a workgroup's tiles must share
// their column tile (workgroups per XCD % column tiles == 0, or one tile per workgroup): the pre-added bias
// fragment stays in 16 registers for the workgroup's lifetime.
#include <stdlib.h>

#include "common.hpp"
#include "gemm_nt.hpp"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

constexpr int TM = 256, TN = 256, BK = 64;
constexpr int HT = 128 * 128;             // half-tile: 128 rows x 64 k bf16 = 16 KB
constexpr int RING_BYTES = 8 * HT;        // [K-tile parity][A0 A1 B0 B1]
constexpr int LDS_BYTES = RING_BYTES + TN * 4;   // + the workgroup's bias slice (fp32, pre-added)

struct Nt256rArgs {
    const bf16_t* A;
    const bf16_t* B;
    bf16_t* C;
    const float* bias1;
    const float* bias2;
    long long lda, ldb, ldc;
    int M, N, K;
    int n_tiles, tiles;
    float2* lse_part;   // optional [M][lse_slots] (max, sum exp(x - max)) over 64-column slots of each C row
    int lse_slots;
    int dbg;            // EDGEDICT_NT256_DEBUG ablation bits (tools/nt256_ablate.py): 1 no lse, 2 no C store, 4 no MFMA,
                        // 16 write-through C stores instead of plain ones, 32 non-temporal C stores, 64 clock probe (cycles / 100 MHz ticks per
                        // workgroup over the first partials)
};

__device__ __forceinline__ void glds16(const bf16_t* src, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <bool LSE>
__global__ __launch_bounds__(512, 1) void gemm_nt256r_kernel(Nt256rArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;        // wave rows {wm*64 + mh*128 ..+64}, wave cols wn*64 ..+64
    const int r16 = lane & 15, kq = lane >> 4;
    const int KT = g.K / BK;
    // VMEM operations of one FULL tile's epilogue per wave: 8 row fragments x 2 C stores (+ 1 partial store)
    constexpr int EPI_OPS = LSE ? 24 : 16;

    // ---- this workgroup's walk: the tiles of XCD x (blockIdx % 8) are one contiguous range; its workgroups take
    // them round-robin, so at any time an XCD works on (about) `wgx` adjacent tiles that share A row panels
    const int nwg = gridDim.x, xcd = blockIdx.x % 8, iw = blockIdx.x / 8;
    const int wgx = (nwg - xcd + 7) / 8;
    const int xq = g.tiles / 8, xr = g.tiles % 8;
    const int x_start = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
    const int x_count = xq + (xcd < xr ? 1 : 0);
    if (iw >= x_count) return;
    const int my_tiles = (x_count - iw + wgx - 1) / wgx;

    // ---- DMA lane constants.  A piece = 8 rows x 128 bytes = one wave instruction; lane l: row l >> 3, LDS
    // chunk l & 7, which holds SOURCE chunk (l & 7) ^ (row & 7).  Wave w brings pieces w and w + 8 of each half-tile.
    // A half h: tile rows h*128 + piece*8 + lr.  B half h: piece p holds tile rows (p >> 2)*64 + h*32 + (p & 3)*8 +
    // lr, i.e. n-wave (p >> 2)'s columns [h*32, h*32 + 32).
    const int lr = lane >> 3;
    const int sch = ((lane & 7) ^ lr) * 8;
    // per lane: 32-bit byte offsets from the tile's (uniform) row-panel bases
    unsigned aO[2][2], bO[2][2];
    const unsigned char* aBase;
    const unsigned char* bBase;
    int i_tile = 0, i_kt = 0, i_par = 0;      // issue cursor: index in the walk, K tile, slot parity
    auto set_issue_tile = [&](int ti) {
        const int tile = x_start + iw + ti * wgx;
        const int m0 = (tile / g.n_tiles) * TM, n0 = (tile % g.n_tiles) * TN;
        aBase = reinterpret_cast<const unsigned char*>(g.A + (long long)m0 * g.lda);
        bBase = reinterpret_cast<const unsigned char*>(g.B + (long long)n0 * g.ldb);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int ar = h * 128 + s * 64 + wave * 8 + lr;
                const int br = ((wave >> 2) + 2 * s) * 64 + h * 32 + (wave & 3) * 8 + lr;
                aO[h][s] = (unsigned)((min(ar, g.M - 1 - m0) * (int)g.lda + sch) * 2);
                bO[h][s] = (unsigned)((min(br, g.N - 1 - n0) * (int)g.ldb + sch) * 2);
            }
    };
    // kind: 0 A0, 1 B0, 2 B1, 3 A1 (issue order inside a K tile); the cursor moves on after A1.  Past the end of the
    // walk the last tile is fetched again into slots nobody reads: the counted waits stay uniform.
    auto issue = [&](int kind) {
        const int slot = kind == 0 ? 0 : kind == 3 ? 1 : kind == 1 ? 2 : 3;
        unsigned char* base = smem + (i_par * 4 + slot) * HT;
        const int h = (kind == 2 || kind == 3) ? 1 : 0;
        const bool isA = (kind == 0 || kind == 3);
        const unsigned char* src = (isA ? aBase : bBase) + i_kt * (BK * 2);
#pragma unroll
        for (int s = 0; s < 2; ++s)
            glds16(reinterpret_cast<const bf16_t*>(src + (isA ? aO[h][s] : bO[h][s])), base + (wave + 8 * s) * 1024);
        if (kind == 3) {
            i_par ^= 1;
            if (++i_kt == KT) {
                i_kt = 0;
                if (i_tile + 1 < my_tiles) set_issue_tile(++i_tile);
            }
        }
    };

    // ---- the workgroup's bias slice (pre-added) goes to LDS once (see the requirements); the accumulators start
    // from it at every tile
    float* sBias = reinterpret_cast<float*>(smem + RING_BYTES);
    if (threadIdx.x < TN) {
        const int nc = ((x_start + iw) % g.n_tiles) * TN + threadIdx.x;
        float v = 0.f;
        if (nc < g.N) v = (g.bias1 ? g.bias1[nc] : 0.f) + (g.bias2 ? g.bias2[nc] : 0.f);
        sBias[threadIdx.x] = v;
    }
    set_issue_tile(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const float* myBias = sBias + wn * 64 + kq * 4;     // + j * 16

    f32x4_t acc[8][4];       // [mh*4 + i][nh*2 + j]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4_t bj = *reinterpret_cast<const f32x4_t*>(myBias + j * 16);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i][j] = bj;
    }

    // fragment read offsets inside a half-tile slot (k-step 1 = offset ^ 64)
    const int a_off = (wm * 64 + r16) * 128 + ((kq ^ (r16 & 7)) << 4);     // + i * 2048
    const int b_off = (wn * 32 + r16) * 128 + ((kq ^ (r16 & 7)) << 4);     // + j * 2048

    // ---- prologue: K tile 0 complete + A0, B0 of K tile 1
    issue(0); issue(1); issue(2); issue(3);
    issue(0); issue(1);
    wait_vm<8>();
    asm volatile("s_barrier" ::: "memory");
    if (wm == 1) asm volatile("s_barrier" ::: "memory");

    bf16x8_t a[4][2], b0[2][2], b1[2][2];
    int c_par = 0;

    // One K tile = 4 phases [LOAD | barrier | 16 MFMA | barrier].  XTRA = VMEM operations (epilogue stores) this wave
    // issued between the DMA a wait retires and the wait itself.
    bool cols_live = true;   // this wave owns at least one valid column of the current tile (N = 640: the third column
                             // tile is half empty - its n-waves 2, 3 keep the barriers and leave the matrix pipe alone)
    auto mfma16 = [&](int mh, int nh, bf16x8_t (&bb)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
        if (!(g.dbg & 4) && cols_live) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[mh * 4 + i][nh * 2 + j] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb[j][ks], a[i][ks], acc[mh * 4 + i][nh * 2 + j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto read_a = [&](const unsigned char* s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i][0] = *reinterpret_cast<const bf16x8_t*>(s + a_off + i * 2048);
            a[i][1] = *reinterpret_cast<const bf16x8_t*>(s + (a_off ^ 64) + i * 2048);
        }
    };
    auto read_b = [&](const unsigned char* s, bf16x8_t (&bb)[2][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bb[j][0] = *reinterpret_cast<const bf16x8_t*>(s + b_off + j * 2048);
            bb[j][1] = *reinterpret_cast<const bf16x8_t*>(s + (b_off ^ 64) + j * 2048);
        }
    };
#define ED_WAIT(LAX) do { if (LAX) wait_vm<8 + EPI_OPS>(); else wait_vm<8>(); } while (0)
#define ED_KTILE(XTRA)                                                                                     \
    {                                                                                                      \
        const unsigned char* buf = smem + c_par * 4 * HT;                                                  \
        /* phase 0: (m0, n0) */                                                                            \
        read_b(buf + 2 * HT, b0);                                                                          \
        read_a(buf);                                                                                       \
        issue(2);                                                                                          \
        ED_WAIT(XTRA);            /* retires B1 of this K tile */                                    \
        asm volatile("s_barrier" ::: "memory");                                                            \
        mfma16(0, 0, b0);                                                                                  \
        asm volatile("s_barrier" ::: "memory");                                                            \
        /* phase 1: (m0, n1) */                                                                            \
        read_b(buf + 3 * HT, b1);                                                                          \
        issue(3);                                                                                          \
        ED_WAIT(XTRA);            /* retires A1 of this K tile */                                    \
        asm volatile("s_barrier" ::: "memory");                                                            \
        mfma16(0, 1, b1);                                                                                  \
        asm volatile("s_barrier" ::: "memory");                                                            \
        /* phase 2: (m1, n1) */                                                                            \
        read_a(buf + HT);                                                                                  \
        issue(0);                                                                                          \
        asm volatile("s_barrier" ::: "memory");                                                            \
        mfma16(1, 1, b1);                                                                                  \
        asm volatile("s_barrier" ::: "memory");                                                            \
        /* phase 3: (m1, n0), B0 still in registers */                                                     \
        issue(1);                                                                                          \
        ED_WAIT(XTRA);            /* retires A0, B0 of the next K tile */                            \
        asm volatile("s_barrier" ::: "memory");                                                            \
        mfma16(1, 0, b0);                                                                                  \
        asm volatile("s_barrier" ::: "memory");                                                            \
        c_par ^= 1;                                                                                        \
    }

    const long long dbg_c0 = (g.dbg & 64) ? clock64() : 0, dbg_w0 = (g.dbg & 64) ? wall_clock64() : 0;
    bool lax = false;        // the previous tile's epilogue issued exactly EPI_OPS VMEM operations
    for (int ti = 0; ti < my_tiles; ++ti) {
        const int tile = x_start + iw + ti * wgx;
        const int m0 = (tile / g.n_tiles) * TM, n0 = (tile % g.n_tiles) * TN;
        cols_live = n0 + wn * 64 < g.N;
        ED_KTILE(lax)
        for (int kt = 1; kt < KT; ++kt) ED_KTILE(false)

        // ---- epilogue, straight from the accumulators (the ring keeps filling)
        const bool full = (m0 + TM <= g.M) && (n0 + TN <= g.N);
        constexpr float LOG2E = 1.4426950408889634f;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int row = m0 + (mi >> 2) * 128 + wm * 64 + (mi & 3) * 16 + r16;
            if (LSE && !(g.dbg & 1)) {
                // per C row, (max, sum exp(x - max)) over this wave's 64 columns, from the bf16-ROUNDED values
                float v[16];
                float mx = -INFINITY;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int nc = n0 + wn * 64 + j * 16 + kq * 4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float x = bf16_to_f32(f32_to_bf16(acc[mi][j][q]));
                        if (!full && nc + q >= g.N) x = -INFINITY;
                        v[j * 4 + q] = x;
                        mx = fmaxf(mx, x);
                    }
                }
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
                const int slot = (n0 >> 6) + wn;
                float2* dst = g.lse_part + (long long)min(row, g.M - 1) * g.lse_slots + min(slot, g.lse_slots - 1);
                if (full) {
                    if (kq == 0) *dst = make_float2(mx, sm);
                } else if (kq == 0 && row < g.M && slot < g.lse_slots) {
                    *dst = make_float2(mx, sm);
                }
            }
            // pack, pair lanes kq <-> kq ^ 1: even kq ends up with 8 consecutive columns of fragment 2 jp, odd kq with
            // those of fragment 2 jp + 1
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                unsigned x0 = f32x2_to_bf16x2(acc[mi][2 * jp][0], acc[mi][2 * jp][1]);
                unsigned x1 = f32x2_to_bf16x2(acc[mi][2 * jp][2], acc[mi][2 * jp][3]);
                unsigned y0 = f32x2_to_bf16x2(acc[mi][2 * jp + 1][0], acc[mi][2 * jp + 1][1]);
                unsigned y1 = f32x2_to_bf16x2(acc[mi][2 * jp + 1][2], acc[mi][2 * jp + 1][3]);
                auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                const u32x4_t v = {s0[0], s1[0], s0[1], s1[1]};
                const int col = n0 + wn * 64 + jp * 32 + (kq & 1) * 16 + (kq >> 1) * 8;
                bf16_t* dst = g.C + (long long)row * g.ldc + col;
                if (g.dbg & 2) {
                } else if (full) {
                    // (plain stores: with 128-byte row segments leaving straight from the registers the write-through
                    // form that helped the LDS-staged epilogue of gemm_nt256.hip is 5 % SLOWER here - 31.8 vs 30.2 us per
                    // logits tile, profiles/r6_nt256r_ablate.txt)
                    if (g.dbg & 16) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
                    else if (g.dbg & 32) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(v) : "memory");
                    else *reinterpret_cast<u32x4_t*>(dst) = v;
                } else if (row < g.M && col < g.N) {
                    *reinterpret_cast<u32x4_t*>(dst) = v;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[mi][j] = *reinterpret_cast<const f32x4_t*>(myBias + j * 16);
        }
        lax = full && !(g.dbg & 3);
    }
#undef ED_KTILE
#undef ED_WAIT
    if (wm == 0) asm volatile("s_barrier" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((g.dbg & 64) && LSE && threadIdx.x == 0) {
        // probe: shader cycles and 100 MHz ticks of this workgroup's whole walk (overwrites the first partials)
        long long* o = reinterpret_cast<long long*>(g.lse_part) + 2 * blockIdx.x;
        o[0] = clock64() - dbg_c0;
        o[1] = wall_clock64() - dbg_w0;
    }
}

}  // namespace

bool ed_gemm_nt256r_ok(int M, int N, int K, bool has_bias, int* grid_out) {
    // (read at every call: tests/test_gemm_gpu.py compares both kernels in one process)
    const char* e = getenv("EDGEDICT_GEMM_NT256R");
    if ((e && atoi(e) == 0) || !ed_gemm_nt256_shape_ok(M, N, K)) return false;
    const int n_tiles = (N + TN - 1) / TN;
    const long long tiles = (long long)((M + TM - 1) / TM) * n_tiles;
    if (tiles >= (1ll << 31)) return false;
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            n = 256;
        return n > 0 ? n - n % 8 : 256;
    }();
    const int grid = (int)(tiles < cus ? tiles : cus);
    // a workgroup keeps ONE bias fragment: all of its tiles must lie in the same column tile
    if (has_bias && tiles > grid && ((grid / 8) % n_tiles != 0 || grid % 8 != 0)) return false;
    if (grid_out) *grid_out = grid;
    return true;
}

int ed_gemm_nt256r_launch(const void* A, long long lda, const void* B, long long ldb, void* C, long long ldc, int M,
                          int N, int K, const float* bias1, const float* bias2, hipStream_t s, float* lse_part) {
    int grid = 0;
    ED_CHECK_ARG(ed_gemm_nt256r_ok(M, N, K, bias1 || bias2, &grid), "gemm_nt256r: shape not covered");
    Nt256rArgs g;
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = (bf16_t*)C;
    g.bias1 = bias1; g.bias2 = bias2;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K;
    g.lse_part = (float2*)lse_part;
    g.lse_slots = (N + 63) / 64;
    g.n_tiles = (N + TN - 1) / TN;
    g.tiles = ((M + TM - 1) / TM) * g.n_tiles;
    static const int dbg = [] { const char* e = getenv("EDGEDICT_NT256_DEBUG"); return e ? atoi(e) : 0; }();
    g.dbg = dbg;
    if (lse_part) {
        ED_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_nt256r_kernel<true>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        hipLaunchKernelGGL(gemm_nt256r_kernel<true>, dim3((unsigned)grid), dim3(512), LDS_BYTES, s, g);
    } else {
        ED_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_nt256r_kernel<false>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        hipLaunchKernelGGL(gemm_nt256r_kernel<false>, dim3((unsigned)grid), dim3(512), LDS_BYTES, s, g);
    }
    ED_CHECK_LAUNCH("gemm_nt256r");
    return ED_OK;
}
