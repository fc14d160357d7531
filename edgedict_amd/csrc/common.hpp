// Shared device/host helpers for the edgedict_amd HIP kernels (gfx950 / CDNA4 only).
//
// Conventions used by every translation unit in this directory:
//   * one wavefront = 64 lanes (hard-coded, see cdna_hip_programming.md §1);
//   * every entry point is extern "C", takes raw device pointers + sizes + a hipStream_t,
//     allocates nothing, and returns an int status (0 = ok);
//   * "dtype" arguments use the ED_F32 / ED_BF16 codes from include/edgedict_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>

#include "edgedict_hip.h"

#define ED_WAVE 64

typedef unsigned short bf16_t;  // raw bf16 bits; arithmetic is always done in fp32

// thread-local last-error text, exported through edgedict_last_error()
void ed_set_error(const char* fmt, ...);

#define ED_CHECK_ARG(cond, ...)                       \
    do {                                              \
        if (!(cond)) {                                \
            ed_set_error(__VA_ARGS__);                \
            return ED_ERR_INVALID;                    \
        }                                             \
    } while (0)

#define ED_CHECK_LAUNCH(name)                                              \
    do {                                                                   \
        hipError_t e__ = hipGetLastError();                                \
        if (e__ != hipSuccess) {                                           \
            ed_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return ED_ERR_LAUNCH;                                          \
        }                                                                  \
    } while (0)

#define ED_CHECK_HIP(expr)                                                 \
    do {                                                                   \
        hipError_t e__ = (expr);                                           \
        if (e__ != hipSuccess) {                                           \
            ed_set_error("%s failed: %s", #expr, hipGetErrorString(e__));  \
            return ED_ERR_LAUNCH;                                          \
        }                                                                  \
    } while (0)

// ---------------------------------------------------------------- bf16 <-> f32
__device__ __forceinline__ float bf16_to_f32(bf16_t b) {
    return __uint_as_float(((unsigned)b) << 16);
}
// round-to-nearest-even, NaN preserved (same rounding torch uses for .to(bfloat16))
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    // gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, round-to-nearest-even); the bit
    // arithmetic it replaces was ~6 VALU instructions per element - a third of rnnt_grad's work
    const __bf16 b = (__bf16)f;
    return *reinterpret_cast<const bf16_t*>(&b);
}
// two values -> one packed 32-bit word (low half = a), ONE instruction
__device__ __forceinline__ unsigned f32x2_to_bf16x2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
    const f32x2_ v = {a, b};
    const bf16x2_ r = __builtin_convertvector(v, bf16x2_);
    return *reinterpret_cast<const unsigned*>(&r);
}

template <typename T> struct ElemIO;
template <> struct ElemIO<float> {
    static constexpr int VEC = 4;  // elements per 16-byte access
    __device__ static __forceinline__ float load(const float* p) { return *p; }
    __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
    __device__ static __forceinline__ void load_vec(const float* p, float (&o)[4]) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    __device__ static __forceinline__ void store_vec(float* p, const float (&o)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    }
    // the 16 raw bytes now, the floats later (lets a kernel put independent work between request and use)
    __device__ static __forceinline__ void cvt_vec(const uint4& v, float (&o)[4]) {
        o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w);
    }
};
template <> struct ElemIO<bf16_t> {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
    __device__ static __forceinline__ void load_vec(const bf16_t* p, float (&o)[8]) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = __uint_as_float(w[i] << 16);
            o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    __device__ static __forceinline__ void cvt_vec(const uint4& v, float (&o)[8]) {
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = __uint_as_float(w[i] << 16);
            o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    __device__ static __forceinline__ void store_vec(bf16_t* p, const float (&o)[8]) {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[i] = f32x2_to_bf16x2(o[2 * i], o[2 * i + 1]);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};

// ---------------------------------------------------------------- wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// log(exp(a)+exp(b)) that tolerates -inf on either side
__device__ __forceinline__ float log_add(float a, float b) {
    const float m = fmaxf(a, b);
    if (m == -INFINITY) return -INFINITY;
    return m + log1pf(expf(-fabsf(a - b)));
}

// Counter-based dropout mask (elementwise.hip dropout_kernel, the encoder stack's norm role and LayerNorm backward):
// element i of the tensor is KEPT iff hash(seed, i) >= p * 2^32; stateless, so the backward pass regenerates it.
__device__ __forceinline__ unsigned ed_drop_hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ bool ed_drop_keep(unsigned seed, long long i, unsigned thresh) {
    return ed_drop_hash(seed ^ ed_drop_hash((unsigned)i * 0x9e3779b9U + (unsigned)(i >> 32))) >= thresh;
}
__host__ __device__ static inline unsigned ed_drop_thresh(float p) { return (unsigned)((double)p * 4294967296.0); }

static inline int ed_grid_for(long long work_items, int per_block, int max_blocks = 256 * 8) {
    long long g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}
