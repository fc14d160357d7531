// MFMA GEMM for gfx950:  C[M,N] (+)= A[M,K] * B[N,K]^T + bias1[N] + bias2[N]
//
// One kernel template serves every dense product on the RNN-T path (reference call sites:
// nn.LSTM input products rnnt/models.py:45-46,65; nn.Linear at rnnt/models.py:129,135,148,156,
// 165-167 and their autograd transposes).  Both operands are described by (pointer, leading
// dimension, k_major flag):
//     k_major = 1 : element (row, k) at  p[row*ld + k]   (K contiguous: x @ W^T forward form)
//     k_major = 0 : element (row, k) at  p[k*ld + row]   (row contiguous: the transposed
//                   operand of dX = dY*W and dW = dY^T*X; transposed on the way into LDS)
// so NT / NN / TN / TT products need no materialised transposes.
//
// Tiling (wave64, 256 threads = 2x2 waves): block tile 128x128, wave tile 64x64 = 4x4 MFMA tiles
// of 16x16.  bf16 inputs use v_mfma_f32_16x16x32_bf16 (BK = 64), fp32 inputs use the exact
// v_mfma_f32_16x16x4_f32 (BK = 16).  Accumulation is always fp32.  Operand tiles are staged
// global -> registers -> LDS (rows padded by 16 B against ds_read bank conflicts) with the next
// tile's global loads issued before the current tile's MFMAs.
// split_k > 1 partitions K over workgroups (XCD-aware: one K slice per XCD) and combines with fp32
// atomics (gradient "+=").
#include <stdlib.h>

#include "common.hpp"
#include "gemm_nt.hpp"
#include "blaslt.hpp"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int BM = 128, BN = 128;
constexpr int THREADS = 256;

template <typename TI> struct Cfg;
template <> struct Cfg<bf16_t> {
    static constexpr int BK = 64;      // elements of K per LDS tile
    static constexpr int ROW = BK + 8; // padded LDS row (elements): 144 B
    static constexpr int VEC = 8;      // elements per 16-byte chunk
};
template <> struct Cfg<float> {
    static constexpr int BK = 16;
    static constexpr int ROW = BK + 4;  // 80 B
    static constexpr int VEC = 4;
};

struct GemmArgs {
    const void* A;
    const void* B;
    void* C;
    const float* bias1;
    const float* bias2;
    long long lda, ldb, ldc;
    int M, N, K;
    int a_vec, b_vec;  // 16-byte vector path usable for the operand
    int c_vec;         // bf16 output rows are 16-byte addressable (ldc % 8 == 0, N % 8 == 0)
    int accumulate;
    int split_k;
    int k_per_split;  // multiple of BK
    int items;        // tiles * split_k
    float* partials;  // non-null: slice s stores its tiles to partials + s*partial_stride ([M][N] dense,
    long long partial_stride;   // plain stores, no atomics); the caller sums the slices
};

// ---- staging registers: the 16-byte chunks one thread moves per operand tile
template <typename TI> struct Stage { uint4 v[Cfg<TI>::BK * BM / Cfg<TI>::VEC / THREADS]; };

template <typename TI, bool FAST>
__device__ __forceinline__ uint4 load_chunk_guarded(const TI* p, long long ld, int vec_ok,
                                                    int r0, int rmax, int c0, int cmax,
                                                    bool along_row) {
    // Loads VEC elements starting at logical (r0, c0) walking along the contiguous dimension.
    // along_row: contiguous index is c (k_major: r = row, c = k); else contiguous index is r.
    constexpr int VEC = Cfg<TI>::VEC;
    if constexpr (FAST) {
        // host guarantees 16-byte alignment and that the contiguous extent is a multiple of VEC,
        // so a chunk is either wholly inside or wholly outside: one unconditional load from a
        // clamped address + select.  No branches, no waits: every load of a tile is in flight
        // together.
        const bool ok = (r0 < rmax) && (c0 < cmax);
        const long long off = along_row ? (long long)r0 * ld + c0 : (long long)c0 * ld + r0;
        uint4 v = *reinterpret_cast<const uint4*>(p + (ok ? off : 0));
        if (!ok) v = make_uint4(0, 0, 0, 0);
        return v;
    }
    uint4 out = make_uint4(0, 0, 0, 0);
    TI* o = reinterpret_cast<TI*>(&out);
    if (along_row) {
        if (r0 >= rmax) return out;
        const TI* src = p + (long long)r0 * ld + c0;
        if (vec_ok && c0 + VEC <= cmax) return *reinterpret_cast<const uint4*>(src);
#pragma unroll
        for (int i = 0; i < VEC; ++i)
            if (c0 + i < cmax) o[i] = src[i];
    } else {
        if (c0 >= cmax) return out;
        const TI* src = p + (long long)c0 * ld + r0;
        if (vec_ok && r0 + VEC <= rmax) return *reinterpret_cast<const uint4*>(src);
#pragma unroll
        for (int i = 0; i < VEC; ++i)
            if (r0 + i < rmax) o[i] = src[i];
    }
    return out;
}

// K-major operand: chunk c -> (row = c / (BK/VEC), kc = c % (BK/VEC))
template <typename TI, bool FAST>
__device__ __forceinline__ void gload_kmajor(Stage<TI>& st, const TI* p, long long ld, int vec_ok,
                                             int row0, int rows, int k0, int kend) {
    constexpr int CPR = Cfg<TI>::BK / Cfg<TI>::VEC;
    constexpr int N = Cfg<TI>::BK * BM / Cfg<TI>::VEC / THREADS;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int c = threadIdx.x + i * THREADS;
        const int r = c / CPR, kc = c % CPR;
        st.v[i] = load_chunk_guarded<TI, FAST>(p, ld, vec_ok, row0 + r, rows,
                                               k0 + kc * Cfg<TI>::VEC, kend, true);
    }
}
// bf16 tiles: the 16-byte chunk kc of row r lives at chunk position kc ^ ((r >> 3) & 7).  Rows that
// are 8 apart start on the same LDS bank (8 * 144 B = 9 * 128 B), which is exactly the stride of
// the transposing store below; the XOR spreads them over the 8 chunk positions.
template <typename TI> __device__ __forceinline__ int swz(int r, int kc) { return kc; }
template <> __device__ __forceinline__ int swz<bf16_t>(int r, int kc) { return kc ^ ((r >> 3) & 7); }

template <typename TI>
__device__ __forceinline__ void sstore_kmajor(const Stage<TI>& st, TI* lds) {
    constexpr int CPR = Cfg<TI>::BK / Cfg<TI>::VEC;
    constexpr int N = Cfg<TI>::BK * BM / Cfg<TI>::VEC / THREADS;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int c = threadIdx.x + i * THREADS;
        const int r = c / CPR, kc = c % CPR;
        *reinterpret_cast<uint4*>(lds + r * Cfg<TI>::ROW + swz<TI>(r, kc) * Cfg<TI>::VEC) = st.v[i];
    }
}

// Row-major-in-k ("transposed") operand.  fp32: item = (k, 4 rows); bf16: item = (k pair, 8 rows)
template <bool FAST>
__device__ __forceinline__ void gload_tr(Stage<float>& st, const float* p, long long ld,
                                         int vec_ok, int row0, int rows, int k0, int kend) {
    constexpr int CPK = BM / 4;  // chunks per k line
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = threadIdx.x + i * THREADS;
        const int k = c / CPK, rc = c % CPK;
        st.v[i] = load_chunk_guarded<float, FAST>(p, ld, vec_ok, row0 + rc * 4, rows, k0 + k, kend,
                                                  false);
    }
}
__device__ __forceinline__ void sstore_tr(const Stage<float>& st, float* lds) {
    constexpr int CPK = BM / 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = threadIdx.x + i * THREADS;
        const int k = c / CPK, rc = c % CPK;
        const float* f = reinterpret_cast<const float*>(&st.v[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) lds[(rc * 4 + j) * Cfg<float>::ROW + k] = f[j];
    }
}
template <bool FAST>
__device__ __forceinline__ void gload_tr(Stage<bf16_t>& st, const bf16_t* p, long long ld,
                                         int vec_ok, int row0, int rows, int k0, int kend) {
    constexpr int CPK = BM / 8;  // 16 chunks per k line
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int item = threadIdx.x + i * THREADS;  // 512 items = 32 k-pairs x 16 chunks
        const int kp = item / CPK, rc = item % CPK;
        st.v[2 * i] = load_chunk_guarded<bf16_t, FAST>(p, ld, vec_ok, row0 + rc * 8, rows,
                                                       k0 + 2 * kp, kend, false);
        st.v[2 * i + 1] = load_chunk_guarded<bf16_t, FAST>(p, ld, vec_ok, row0 + rc * 8, rows,
                                                           k0 + 2 * kp + 1, kend, false);
    }
}
__device__ __forceinline__ void sstore_tr(const Stage<bf16_t>& st, bf16_t* lds) {
    constexpr int CPK = BM / 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int item = threadIdx.x + i * THREADS;
        const int kp = item / CPK, rc = item % CPK;
        const unsigned* a = reinterpret_cast<const unsigned*>(&st.v[2 * i]);      // k even
        const unsigned* b = reinterpret_cast<const unsigned*>(&st.v[2 * i + 1]);  // k odd
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // rows 2j, 2j+1 of this chunk; pack (k, k+1) for each row into one dword
            const unsigned lo = (a[j] & 0xffffu) | (b[j] << 16);
            const unsigned hi = (a[j] >> 16) | (b[j] & 0xffff0000u);
            // all 8 rows of this chunk share (row >> 3) = rc, hence one swizzled chunk position
            const int pos = (((kp >> 2) ^ (rc & 7)) << 3) + ((2 * kp) & 7);
            unsigned* d0 = reinterpret_cast<unsigned*>(lds + (rc * 8 + 2 * j) * Cfg<bf16_t>::ROW + pos);
            unsigned* d1 = reinterpret_cast<unsigned*>(lds + (rc * 8 + 2 * j + 1) * Cfg<bf16_t>::ROW + pos);
            *d0 = lo;
            *d1 = hi;
        }
    }
}

template <typename TI, bool KMAJOR, bool FAST>
__device__ __forceinline__ void gload(Stage<TI>& st, const TI* p, long long ld, int vec_ok,
                                      int row0, int rows, int k0, int kend) {
    if constexpr (KMAJOR) gload_kmajor<TI, FAST>(st, p, ld, vec_ok, row0, rows, k0, kend);
    else gload_tr<FAST>(st, p, ld, vec_ok, row0, rows, k0, kend);
}
template <typename TI, bool KMAJOR>
__device__ __forceinline__ void sstore(const Stage<TI>& st, TI* lds) {
    if constexpr (KMAJOR) sstore_kmajor<TI>(st, lds);
    else sstore_tr(st, lds);
}

// ---- MFMA over one LDS tile pair
__device__ __forceinline__ void mma_tile(const bf16_t* sA, const bf16_t* sB, f32x4_t (&acc)[4][4],
                                         int wm, int wn, int lane) {
    constexpr int ROW = Cfg<bf16_t>::ROW;
    const int r = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < Cfg<bf16_t>::BK / 32; ++ks) {
        bf16x8_t a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ra = wm * 64 + i * 16 + r, rb = wn * 64 + i * 16 + r;
            a[i] = *reinterpret_cast<const bf16x8_t*>(sA + ra * ROW + swz<bf16_t>(ra, ks * 4 + kq) * 8);
            b[i] = *reinterpret_cast<const bf16x8_t*>(sB + rb * ROW + swz<bf16_t>(rb, ks * 4 + kq) * 8);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}
__device__ __forceinline__ void mma_tile(const float* sA, const float* sB, f32x4_t (&acc)[4][4],
                                         int wm, int wn, int lane) {
    constexpr int ROW = Cfg<float>::ROW;
    const int r = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < Cfg<float>::BK / 4; ++ks) {
        float a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = sA[(wm * 64 + i * 16 + r) * ROW + ks * 4 + kq];
            b[i] = sB[(wn * 64 + i * 16 + r) * ROW + ks * 4 + kq];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}

template <typename TI, typename TO, bool A_KM, bool B_KM, bool FAST>
__global__ __launch_bounds__(THREADS) void gemm_kernel(GemmArgs g) {
    using C_ = Cfg<TI>;
    __shared__ __attribute__((aligned(16))) TI smem[(BM + BN) * C_::ROW];
    TI* sA = smem;
    TI* sB = smem + BM * C_::ROW;

    // N index fastest: consecutive blocks share the same A row panel (L2 reuse)
    const int n_tiles = (g.N + BN - 1) / BN;
    // split-K launches are 1-D with the K slice FASTEST (slice = id % split_k, split_k % 8 == 0):
    // workgroup id % 8 is the XCD, so every tile of one K slice runs on the SAME XCD and the slice's
    // operand rows are fetched from HBM once into that XCD's L2 instead of once per XCD
    // (measured on the joint dW2 product: 26.5 GB of HBM reads for 4.5 GB of operands before).
    // Work items = tiles x K slices.  A normal launch has one workgroup per item; a BACKGROUND launch
    // (edgedict_gemm_bg) has only as many workgroups as are resident at once and each walks its
    // items (stride gridDim.x, a multiple of split_k: the slice, hence the XCD, stays fixed).  A
    // grid with more workgroups than fit parks its tail in the dispatcher, and on this chip that
    // blocks the dispatch of OTHER queues' kernels (measured: 60 us per tiny kernel on the main
    // stream while a 1280-workgroup dW GEMM ran on the auxiliary stream).
    const TI* A = reinterpret_cast<const TI*>(g.A);
    const TI* B = reinterpret_cast<const TI*>(g.B);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
  for (int item = blockIdx.x; item < g.items; item += gridDim.x) {
    const int tile = g.split_k > 1 ? item / g.split_k : item;
    const int slice = g.split_k > 1 ? item % g.split_k : 0;
    const int m0 = (tile / n_tiles) * BM;
    const int n0 = (tile % n_tiles) * BN;
    const int kbeg = slice * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    Stage<TI> ra, rb;
    if (kbeg < kend) {
        gload<TI, A_KM, FAST>(ra, A, g.lda, g.a_vec, m0, g.M, kbeg, kend);
        gload<TI, B_KM, FAST>(rb, B, g.ldb, g.b_vec, n0, g.N, kbeg, kend);
        sstore<TI, A_KM>(ra, sA);
        sstore<TI, B_KM>(rb, sB);
    }
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += C_::BK) {
        const bool more = (k0 + C_::BK) < kend;
        if (more) {
            gload<TI, A_KM, FAST>(ra, A, g.lda, g.a_vec, m0, g.M, k0 + C_::BK, kend);
            gload<TI, B_KM, FAST>(rb, B, g.ldb, g.b_vec, n0, g.N, k0 + C_::BK, kend);
        }
        mma_tile(sA, sB, acc, wm, wn, lane);
        __syncthreads();
        if (more) {
            sstore<TI, A_KM>(ra, sA);
            sstore<TI, B_KM>(rb, sB);
        }
        __syncthreads();
    }

    // epilogue: lane holds D[row = (lane>>4)*4 + r][col = lane&15] of each 16x16 tile
    TO* C = reinterpret_cast<TO*>(g.C);
    const bool first_split = (slice == 0);
    if constexpr (sizeof(TO) == 2 && sizeof(TI) == 2) {
        if (g.c_vec) {
            // bf16 output: round in registers, stage the 128x128 tile in LDS (the operand tiles
            // are dead after the last barrier) and leave the CU as full 16-byte row segments
            constexpr int CROW = BN + 8;
            bf16_t* sC = reinterpret_cast<bf16_t*>(smem);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cl = wn * 64 + j * 16 + (lane & 15);
                const int col = n0 + cl;
                float bias = 0.f;
                if (col < g.N) {
                    if (g.bias1) bias += g.bias1[col];
                    if (g.bias2) bias += g.bias2[col];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        sC[(wm * 64 + i * 16 + (lane >> 4) * 4 + r) * CROW + cl] =
                            f32_to_bf16(acc[i][j][r] + bias);
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < BM * BN / 8 / THREADS; ++it) {
                const int c = threadIdx.x + it * THREADS;
                const int rl = c / (BN / 8), ch = c % (BN / 8);
                const int row = m0 + rl, col = n0 + ch * 8;
                if (row >= g.M || col >= g.N) continue;
                uint4 v = *reinterpret_cast<const uint4*>(sC + rl * CROW + ch * 8);
                bf16_t* dst = reinterpret_cast<bf16_t*>(C) + (long long)row * g.ldc + col;
                if (g.accumulate) {
                    float x[8], y[8];
                    ElemIO<bf16_t>::load_vec(dst, x);
                    ElemIO<bf16_t>::load_vec(reinterpret_cast<const bf16_t*>(&v), y);
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] += y[e];
                    ElemIO<bf16_t>::store_vec(dst, x);
                } else {
                    *reinterpret_cast<uint4*>(dst) = v;
                }
            }
            __syncthreads();   // staging tile consumed before the next item's operand tiles land
            continue;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 64 + j * 16 + (lane & 15);
        if (col >= g.N) continue;
        float bias = 0.f;
        if (first_split) {
            if (g.bias1) bias += g.bias1[col];
            if (g.bias2) bias += g.bias2[col];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
                if (row >= g.M) continue;
                const float v = acc[i][j][r] + bias;
                TO* dst = C + (long long)row * g.ldc + col;
                if constexpr (sizeof(TO) == 4) {
                    if (g.partials) g.partials[slice * g.partial_stride + (long long)row * g.N + col] = v;
                    else if (g.split_k > 1) atomicAdd(reinterpret_cast<float*>(dst), v);
                    else if (g.accumulate) *dst = *dst + v;
                    else *dst = v;
                } else {
                    if (g.accumulate) ElemIO<TO>::store(dst, ElemIO<TO>::load(dst) + v);
                    else ElemIO<TO>::store(dst, v);
                }
            }
        }
    }
  }   // items
}

template <typename TI, typename TO, bool FAST>
int launch2(const GemmArgs& g, int a_km, int b_km, dim3 grid, hipStream_t s, int pad) {
    // pad = extra dynamic LDS claimed per workgroup (occupancy cap of background GEMMs)
    if (a_km && b_km) hipLaunchKernelGGL((gemm_kernel<TI, TO, true, true, FAST>), grid, dim3(THREADS), pad, s, g);
    else if (a_km && !b_km) hipLaunchKernelGGL((gemm_kernel<TI, TO, true, false, FAST>), grid, dim3(THREADS), pad, s, g);
    else if (!a_km && b_km) hipLaunchKernelGGL((gemm_kernel<TI, TO, false, true, FAST>), grid, dim3(THREADS), pad, s, g);
    else hipLaunchKernelGGL((gemm_kernel<TI, TO, false, false, FAST>), grid, dim3(THREADS), pad, s, g);
    ED_CHECK_LAUNCH("gemm");
    return ED_OK;
}
template <typename TI, typename TO>
int launch(const GemmArgs& g, int a_km, int b_km, dim3 grid, hipStream_t s, int pad) {
    // FAST: both operands 16-byte aligned with a leading dimension that is a multiple of the
    // vector width, and the contiguous extent of each operand a multiple of it as well
    constexpr int VEC = Cfg<TI>::VEC;
    const bool a_ext = a_km ? (g.K % VEC == 0) : (g.M % VEC == 0);
    const bool b_ext = b_km ? (g.K % VEC == 0) : (g.N % VEC == 0);
    if (g.a_vec && g.b_vec && a_ext && b_ext) return launch2<TI, TO, true>(g, a_km, b_km, grid, s, pad);
    return launch2<TI, TO, false>(g, a_km, b_km, grid, s, pad);
}

// C[m][n] (+)= sum_s part[s][m][n]
__global__ void reduce_partials_kernel(const float* __restrict__ part, long long stride, int S,
                                       float* __restrict__ C, long long ldc, long long M, int N,
                                       int accumulate) {
    const long long n = M * N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        float a = 0.f;
        for (int s = 0; s < S; ++s) a += part[s * stride + i];
        float* dst = C + (i / N) * ldc + (i % N);
        *dst = accumulate ? *dst + a : a;
    }
}

__global__ void zero_f32(float* p, long long rows, long long cols, long long ld) {
    const long long n = rows * cols;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        p[(i / cols) * ld + (i % cols)] = 0.f;
}

}  // namespace

static int gemm_impl(int dtype_in, int dtype_out, const void* A, long long lda, int a_kmajor,
                     const void* B, long long ldb, int b_kmajor, void* C, long long ldc, int M, int N,
                     int K, const float* bias1, const float* bias2, int accumulate, int split_k,
                     void* stream_, int max_wg_per_cu, float* partials = nullptr, bool reduce = true) {
    ED_CHECK_ARG(dtype_in == ED_F32 || dtype_in == ED_BF16, "gemm: bad input dtype %d", dtype_in);
    ED_CHECK_ARG(dtype_out == ED_F32 || dtype_out == ED_BF16, "gemm: bad output dtype %d", dtype_out);
    ED_CHECK_ARG(!(dtype_in == ED_F32 && dtype_out == ED_BF16), "gemm: fp32 inputs with bf16 output is not supported");
    ED_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "gemm: negative dimension");
    ED_CHECK_ARG(split_k >= 1, "gemm: split_k must be >= 1");
    ED_CHECK_ARG(split_k == 1 || dtype_out == ED_F32, "gemm: split_k > 1 needs an fp32 output (atomic +=)");
    if (M == 0 || N == 0) return ED_OK;
    ED_CHECK_ARG(A && B && C, "gemm: null operand");
    hipStream_t stream = (hipStream_t)stream_;
    // bf16 NT products with K % 64 == 0 take the direct-to-LDS kernel (gemm_nt.hip)
    static const bool nt_enabled = [] {
        const char* e = getenv("EDGEDICT_GEMM_NT");
        return !(e && e[0] == '0');
    }();
    // large bf16 NT products: the 256 x 256-tile kernel (gemm_nt256.hip), ahead of the vendor route
    if (nt_enabled && max_wg_per_cu == 0 && ed_gemm_nt256_ok(M, N, K, accumulate) &&
        ed_gemm_nt_ok(dtype_in, dtype_out, A, lda, a_kmajor, B, ldb, b_kmajor, C, ldc, M, N, K, split_k, bias1, bias2))
        return ed_gemm_nt256_launch(A, lda, B, ldb, C, ldc, M, N, K, bias1, bias2, stream);
    // the one plain product that is large AND output-heavy (short K: the joint's logits) goes to the
    // vendor library when it is there (blaslt.cpp says why); everything else runs here
    if (dtype_in == ED_BF16 && dtype_out == ED_BF16 && a_kmajor && b_kmajor && !accumulate && !bias2 &&
        split_k == 1 && max_wg_per_cu == 0 && K <= 1024 && N >= 1024 && (long long)M * N >= (1ll << 28) &&
        lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 &&
        ed_blaslt_nt_bf16(A, lda, B, ldb, C, ldc, M, N, K, bias1, 0, stream))
        return ED_OK;
    // ... and the long-K, few-tiles products of the encoder stack's backward chain (dX of a chunk:
    // [768..1536 x 1024 x 4096], 19-23 us there vs 32 us here, tools/blas_probe2.py)
    static const bool small_vendor = [] {
        const char* e = getenv("EDGEDICT_BLASLT_SMALL");   // measured: step 27.17 -> 27.01 ms
        return !(e && e[0] == '0');
    }();
    if (small_vendor && dtype_in == ED_BF16 && dtype_out == ED_BF16 && a_kmajor && b_kmajor && !bias1 && !bias2 &&
        split_k == 1 && max_wg_per_cu == 0 && K >= 2048 && M <= 4096 && N <= 2048 && M >= 256 &&
        lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 &&
        ed_blaslt_nt_bf16(A, lda, B, ldb, C, ldc, M, N, K, nullptr, accumulate, stream))
        return ED_OK;
    if (nt_enabled && max_wg_per_cu == 0 &&
        ed_gemm_nt_ok(dtype_in, dtype_out, A, lda, a_kmajor, B, ldb, b_kmajor, C, ldc, M, N, K, split_k, bias1, bias2))
        return ed_gemm_nt_launch(A, lda, B, ldb, C, ldc, M, N, K, bias1, bias2, accumulate, 0, stream);
    const int esz = dtype_in == ED_F32 ? 4 : 2;
    const int vec = 16 / esz;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias1 = bias1; g.bias2 = bias2;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.a_vec = ((uintptr_t)A % 16 == 0) && (lda % vec == 0);
    g.b_vec = ((uintptr_t)B % 16 == 0) && (ldb % vec == 0);
    g.c_vec = (dtype_out == ED_BF16) && ((uintptr_t)C % 16 == 0) && (ldc % 8 == 0) && (N % 8 == 0);
    g.accumulate = accumulate;
    const int bk = dtype_in == ED_F32 ? 16 : 64;
    int ktiles = (K + bk - 1) / bk;
    if (partials) {
        // quiet mode: few long-lived slices (power of two <= 8 keeps slice <-> XCD set fixed)
        int p2 = 1;
        while (p2 < split_k && p2 < 8) p2 *= 2;
        split_k = p2;
    } else if (split_k > 1) {
        split_k = (split_k + 7) / 8 * 8;   // whole K slices per XCD (see the kernel)
    }
    if (split_k > ktiles) split_k = ktiles > 0 ? ktiles : 1;
    g.split_k = split_k;
    g.k_per_split = ((ktiles + split_k - 1) / split_k) * bk;
    if (g.k_per_split == 0) g.k_per_split = bk;
    g.partials = partials;
    g.partial_stride = (long long)M * N;
    if (split_k > 1 && !accumulate && !partials) {
        // atomics need a defined starting value
        const long long n = (long long)M * N;
        hipLaunchKernelGGL(zero_f32, dim3(ed_grid_for(n, 256)), dim3(256), 0, stream, (float*)C,
                           (long long)M, (long long)N, ldc);
        ED_CHECK_LAUNCH("gemm zero");
    }
    const long long tiles = (long long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    ED_CHECK_ARG(tiles < (1ll << 31), "gemm: too many tiles");
    ED_CHECK_ARG(tiles * split_k < (1ll << 31), "gemm: too many workgroups");
    g.items = (int)(tiles * split_k);
    long long nwg = g.items;
    if (max_wg_per_cu > 0) {
        // background: only as many workgroups as are resident at once (see the kernel)
        static const int n_cu = [] {
            int dev = 0, n = 256;
            if (hipGetDevice(&dev) != hipSuccess ||
                hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
                n = 256;
            return n > 0 ? n : 256;
        }();
        long long cap = (long long)max_wg_per_cu * n_cu;
        if (split_k > 1) cap = cap / split_k * split_k;   // keep item % split_k fixed per workgroup
        if (cap >= split_k && nwg > cap) nwg = cap;
    }
    dim3 grid((unsigned)nwg, 1, 1);
    // occupancy cap: claim enough extra LDS that only max_wg_per_cu workgroups fit on a CU
    int pad = 0;
    if (max_wg_per_cu > 0) {
        const int stat = (BM + BN) * (dtype_in == ED_F32 ? Cfg<float>::ROW * 4 : Cfg<bf16_t>::ROW * 2);
        const int want = (160 * 1024) / (max_wg_per_cu + 1) + 1024;   // one more would not fit
        if (want > stat) pad = (want - stat + 255) / 256 * 256;
    }
    int rc;
    if (dtype_in == ED_BF16 && dtype_out == ED_BF16) rc = launch<bf16_t, bf16_t>(g, a_kmajor, b_kmajor, grid, stream, pad);
    else if (dtype_in == ED_BF16 && dtype_out == ED_F32) rc = launch<bf16_t, float>(g, a_kmajor, b_kmajor, grid, stream, pad);
    else rc = launch<float, float>(g, a_kmajor, b_kmajor, grid, stream, pad);
    if (rc != ED_OK || !partials || !reduce) return rc;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(ed_grid_for((long long)M * N, 256, 2048)), dim3(256), 0,
                       stream, partials, g.partial_stride, g.split_k, (float*)C, ldc, (long long)M, N,
                       accumulate);
    ED_CHECK_LAUNCH("gemm reduce_partials");
    return ED_OK;
}

// internal (encoder_stack.hip): quiet background product that leaves the slices UNREDUCED in
// `partials`; returns the number of slices written through *slices
int ed_gemm_quiet_partials(int dtype_in, const void* A, long long lda, int a_kmajor, const void* B,
                           long long ldb, int b_kmajor, int M, int N, int K, int split_k,
                           int max_wg_per_cu, float* partials, int* slices, hipStream_t stream) {
    int p2 = 1;
    while (p2 < split_k && p2 < 8) p2 *= 2;
    // large bf16 weight-gradient products: 256-row tiles pull fewer bytes per flop through the CU fetch
    // path that the recurrence beside them is bound by - the own kernel (gemm_tn256.hip), else the vendor's
    if (dtype_in == ED_BF16 && !a_kmajor && !b_kmajor && (long long)M * N >= (1ll << 18) && K >= 1024 &&
        ed_gemm_tn256_ok(A, lda, B, ldb, M, N, K)) {
        // (every caller's partials buffer holds 8 slices: encoder_stack.hip tmpW)
        const int S = ed_gemm_tn256_slices(M, N, K, 8);
        *slices = S;
        return ed_gemm_tn256_partials(A, lda, B, ldb, partials, M, N, K, S, 0, stream);
    }
    if (dtype_in == ED_BF16 && !a_kmajor && !b_kmajor && (long long)M * N >= (1ll << 18) && K >= 4096 &&
        lda % 8 == 0 && ldb % 8 == 0 && N % 4 == 0 &&
        ed_blaslt_tn_f32(A, lda, B, ldb, partials, N, M, N, K, 0, stream)) {
        *slices = 1;
        return ED_OK;
    }
    const int bk = dtype_in == ED_F32 ? 16 : 64;
    const int ktiles = (K + bk - 1) / bk;
    if (p2 > ktiles) p2 = ktiles > 0 ? ktiles : 1;
    *slices = p2;
    return gemm_impl(dtype_in, ED_F32, A, lda, a_kmajor, B, ldb, b_kmajor, partials, N, M, N, K, nullptr,
                     nullptr, 0, p2, (void*)stream, max_wg_per_cu, partials, false);
}

extern "C" int edgedict_gemm(int dtype_in, int dtype_out, const void* A, long long lda, int a_kmajor,
                             const void* B, long long ldb, int b_kmajor, void* C, long long ldc,
                             int M, int N, int K, const float* bias1, const float* bias2,
                             int accumulate, int split_k, void* stream_) {
    return gemm_impl(dtype_in, dtype_out, A, lda, a_kmajor, B, ldb, b_kmajor, C, ldc, M, N, K, bias1,
                     bias2, accumulate, split_k, stream_, 0);
}

extern "C" int edgedict_gemm_nt_lse(const void* A, long long lda, const void* B, long long ldb, void* C,
                                    long long ldc, int M, int N, int K, const float* bias,
                                    float* lse_part, void* stream_) {
    ED_CHECK_ARG(A && B && C && lse_part, "gemm_nt_lse: null pointer");
    ED_CHECK_ARG(ed_gemm_nt256_shape_ok(M, N, K) &&
                 ed_gemm_nt_ok(ED_BF16, ED_BF16, A, lda, 1, B, ldb, 1, C, ldc, M, N, K, 1, bias, nullptr),
                 "gemm_nt_lse: needs bf16 K-contiguous operands, K %% 64 == 0, K >= 128, N %% 8 == 0, "
                 "16-byte aligned pointers and leading dimensions (M=%d N=%d K=%d)", M, N, K);
    return ed_gemm_nt256_launch(A, lda, B, ldb, C, ldc, M, N, K, bias, nullptr, (hipStream_t)stream_, lse_part);
}

extern "C" int edgedict_gemm_bg(int dtype_in, int dtype_out, const void* A, long long lda,
                                int a_kmajor, const void* B, long long ldb, int b_kmajor, void* C,
                                long long ldc, int M, int N, int K, const float* bias1,
                                const float* bias2, int accumulate, int split_k,
                                int max_wg_per_cu, float* partials, void* stream_) {
    ED_CHECK_ARG(max_wg_per_cu >= 1 && max_wg_per_cu <= 8, "gemm_bg: max_wg_per_cu must be 1..8");
    ED_CHECK_ARG(!partials || dtype_out == ED_F32, "gemm_bg: the quiet (partials) form needs an fp32 output");
    ED_CHECK_ARG(!partials || (!bias1 && !bias2), "gemm_bg: the quiet (partials) form takes no bias");
    // weight gradients (both operands row-major over the reduction) with a partials buffer: own 256 x 128
    // quiet kernel + the reduce pass
    if (partials && dtype_in == ED_BF16 && dtype_out == ED_F32 && !a_kmajor && !b_kmajor && A && B && C &&
        (long long)M * N >= (1ll << 18) && K >= 1024 && ed_gemm_tn256_ok(A, lda, B, ldb, M, N, K)) {
        int p2 = 1;      // the caller's buffer holds the next power of two >= split_k (<= 8) slices
        while (p2 < split_k && p2 < 8) p2 *= 2;
        p2 = ed_gemm_tn256_slices(M, N, K, p2);
        const int rc = ed_gemm_tn256_partials(A, lda, B, ldb, partials, M, N, K, p2, 0, (hipStream_t)stream_);
        if (rc != ED_OK) return rc;
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(ed_grid_for((long long)M * N, 256, 2048)), dim3(256), 0,
                           (hipStream_t)stream_, partials, (long long)M * N, p2, (float*)C, ldc, (long long)M, N,
                           accumulate);
        ED_CHECK_LAUNCH("gemm reduce_partials");
        return ED_OK;
    }
    if (dtype_in == ED_BF16 && dtype_out == ED_F32 && !a_kmajor && !b_kmajor && !bias1 && !bias2 && A && B && C &&
        (long long)M * N >= (1ll << 18) && K >= 4096 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 &&
        ed_blaslt_tn_f32(A, lda, B, ldb, (float*)C, ldc, M, N, K, accumulate, (hipStream_t)stream_))
        return ED_OK;
    return gemm_impl(dtype_in, dtype_out, A, lda, a_kmajor, B, ldb, b_kmajor, C, ldc, M, N, K, bias1,
                     bias2, accumulate, split_k, stream_, max_wg_per_cu, partials, true);
}
