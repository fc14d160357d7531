// LayerNorm kernels fused with the residual add and the encoder's TimeReduction.
//
// Reference arithmetic (rnnt/models.py):
//   Encoder.forward      :124,132  xs = LayerNorm(input_size)(xs)
//   ResLayerNormLSTM     :66-70    xs = xs + lstm(xs)  (layers > 0);  xs = proj(xs)
//                                  proj = Sequential(LayerNorm(H)[, TimeReduction(2)])
//   TimeReduction        :21-29    zero-pad T to even (AFTER the LayerNorm), mean of frame pairs
// LayerNorm = biased variance, eps inside the sqrt, affine (torch.nn.LayerNorm semantics).
//
// Both kernels are HBM-bound streaming kernels: one wave64 per row (or per pair of rows when
// the time reduction is fused), 16-byte loads, fp32 statistics via the two-pass mean /
// centred-variance form (rows are re-read from L1/L2, never from HBM twice).
#include "common.hpp"

namespace {

template <typename T>
__device__ __forceinline__ float row_sum(const T* x, const T* r, int D, int vec_ok, int lane) {
    constexpr int VEC = ElemIO<T>::VEC;
    float s = 0.f;
    if (vec_ok) {
        for (int c = lane * VEC; c < D; c += 64 * VEC) {
            float a[VEC];
            ElemIO<T>::load_vec(x + c, a);
            if (r) {
                float b[VEC];
                ElemIO<T>::load_vec(r + c, b);
#pragma unroll
                for (int i = 0; i < VEC; ++i) a[i] += b[i];
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) s += a[i];
        }
    } else {
        for (int c = lane; c < D; c += 64) s += ElemIO<T>::load(x + c) + (r ? ElemIO<T>::load(r + c) : 0.f);
    }
    return wave_sum(s);
}

template <typename T>
__device__ __forceinline__ float row_sqdev(const T* x, const T* r, float mean, int D, int vec_ok,
                                           int lane) {
    constexpr int VEC = ElemIO<T>::VEC;
    float s = 0.f;
    if (vec_ok) {
        for (int c = lane * VEC; c < D; c += 64 * VEC) {
            float a[VEC];
            ElemIO<T>::load_vec(x + c, a);
            if (r) {
                float b[VEC];
                ElemIO<T>::load_vec(r + c, b);
#pragma unroll
                for (int i = 0; i < VEC; ++i) a[i] += b[i];
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float d = a[i] - mean;
                s += d * d;
            }
        }
    } else {
        for (int c = lane; c < D; c += 64) {
            const float d = ElemIO<T>::load(x + c) + (r ? ElemIO<T>::load(r + c) : 0.f) - mean;
            s += d * d;
        }
    }
    return wave_sum(s);
}

// out row (b, tau) <- mean over k < reduce of LN(x[b, reduce*tau + k] (+ res)), missing frames = 0
template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd(
    const T* __restrict__ x, const T* __restrict__ res, const float* __restrict__ gamma,
    const float* __restrict__ beta, T* __restrict__ y, float* __restrict__ mean_out,
    float* __restrict__ rstd_out, int B, int Tin, int D, int reduce, float eps, int vec_ok) {
    constexpr int VEC = ElemIO<T>::VEC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Tout = (Tin + reduce - 1) / reduce;
    const long long out_rows = (long long)B * Tout;
    const float inv_red = 1.f / (float)reduce;
    for (long long orow = (long long)blockIdx.x * 4 + wave; orow < out_rows;
         orow += (long long)gridDim.x * 4) {
        const int b = (int)(orow / Tout), tau = (int)(orow % Tout);
        float mean[2] = {0.f, 0.f}, rstd[2] = {0.f, 0.f};
        bool have[2] = {false, false};
        for (int k = 0; k < reduce; ++k) {  // reduce is 1 or 2
            const int t = tau * reduce + k;
            if (t >= Tin) break;
            const long long irow = (long long)b * Tin + t;
            const T* xr = x + irow * D;
            const T* rr = res ? res + irow * D : nullptr;
            const float m = row_sum<T>(xr, rr, D, vec_ok, lane) / (float)D;
            const float var = row_sqdev<T>(xr, rr, m, D, vec_ok, lane) / (float)D;
            const float rs = rsqrtf(var + eps);
            mean[k] = m;
            rstd[k] = rs;
            have[k] = true;
            if (lane == 0) {
                mean_out[irow] = m;
                rstd_out[irow] = rs;
            }
        }
        T* yr = y + orow * D;
        const long long irow0 = (long long)b * Tin + (long long)tau * reduce;
        if (vec_ok) {
            for (int c = lane * VEC; c < D; c += 64 * VEC) {
                float o[VEC];
#pragma unroll
                for (int i = 0; i < VEC; ++i) o[i] = 0.f;
                const float4 g0 = *reinterpret_cast<const float4*>(gamma + c);
                const float4 b0 = *reinterpret_cast<const float4*>(beta + c);
                float gm[VEC], bt[VEC];
                gm[0] = g0.x; gm[1] = g0.y; gm[2] = g0.z; gm[3] = g0.w;
                bt[0] = b0.x; bt[1] = b0.y; bt[2] = b0.z; bt[3] = b0.w;
                if constexpr (VEC == 8) {
                    const float4 g1 = *reinterpret_cast<const float4*>(gamma + c + 4);
                    const float4 b1 = *reinterpret_cast<const float4*>(beta + c + 4);
                    gm[4] = g1.x; gm[5] = g1.y; gm[6] = g1.z; gm[7] = g1.w;
                    bt[4] = b1.x; bt[5] = b1.y; bt[6] = b1.z; bt[7] = b1.w;
                }
                for (int k = 0; k < reduce; ++k) {
                    if (!have[k]) break;
                    float a[VEC];
                    ElemIO<T>::load_vec(x + (irow0 + k) * D + c, a);
                    if (res) {
                        float r2[VEC];
                        ElemIO<T>::load_vec(res + (irow0 + k) * D + c, r2);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) a[i] += r2[i];
                    }
#pragma unroll
                    for (int i = 0; i < VEC; ++i) o[i] += (a[i] - mean[k]) * rstd[k] * gm[i] + bt[i];
                }
                if (reduce > 1) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) o[i] *= inv_red;
                }
                ElemIO<T>::store_vec(yr + c, o);
            }
        } else {
            for (int c = lane; c < D; c += 64) {
                float o = 0.f;
                for (int k = 0; k < reduce; ++k) {
                    if (!have[k]) break;
                    float a = ElemIO<T>::load(x + (irow0 + k) * D + c);
                    if (res) a += ElemIO<T>::load(res + (irow0 + k) * D + c);
                    o += (a - mean[k]) * rstd[k] * gamma[c] + beta[c];
                }
                ElemIO<T>::store(yr + c, o * inv_red);
            }
        }
    }
}

// Backward.  One wave per INPUT row (b,t).  dy row = dout[b, t/reduce] / reduce.
//   xhat = (s - mean) * rstd,  g = dy * gamma
//   ds   = rstd * (g - mean_D(g) - xhat * mean_D(g * xhat))
//   dgamma += dy * xhat,  dbeta += dy       (per-wave LDS accumulators, then global atomics)
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd(
    const T* __restrict__ dout, const T* __restrict__ x, const T* __restrict__ res,
    const float* __restrict__ gamma, const float* __restrict__ mean_in,
    const float* __restrict__ rstd_in, T* __restrict__ ds, float* __restrict__ dgamma,
    float* __restrict__ dbeta, int B, int Tin, int D, int reduce) {
    extern __shared__ __attribute__((aligned(16))) float acc[];  // [4 waves][2][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* ag = acc + (size_t)wave * 2 * D;
    float* ab = ag + D;
    for (int c = lane; c < D; c += 64) {
        ag[c] = 0.f;
        ab[c] = 0.f;
    }
    const int Tout = (Tin + reduce - 1) / reduce;
    const long long rows = (long long)B * Tin;
    const float scale = 1.f / (float)reduce;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < rows;
         row += (long long)gridDim.x * 4) {
        const int b = (int)(row / Tin), t = (int)(row % Tin);
        const T* dyr = dout + ((long long)b * Tout + t / reduce) * D;
        const T* xr = x + row * D;
        const T* rr = res ? res + row * D : nullptr;
        const float m = mean_in[row], rs = rstd_in[row];
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < D; c += 64) {
            const float dy = ElemIO<T>::load(dyr + c) * scale;
            const float xv = ElemIO<T>::load(xr + c) + (rr ? ElemIO<T>::load(rr + c) : 0.f);
            const float xh = (xv - m) * rs;
            const float g = dy * gamma[c];
            s1 += g;
            s2 += g * xh;
            ag[c] += dy * xh;
            ab[c] += dy;
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
        T* dr = ds + row * D;
        for (int c = lane; c < D; c += 64) {
            const float dy = ElemIO<T>::load(dyr + c) * scale;
            const float xv = ElemIO<T>::load(xr + c) + (rr ? ElemIO<T>::load(rr + c) : 0.f);
            const float xh = (xv - m) * rs;
            const float g = dy * gamma[c];
            ElemIO<T>::store(dr + c, rs * (g - s1 - xh * s2));
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float g = 0.f, bsum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            g += acc[(size_t)w * 2 * D + c];
            bsum += acc[(size_t)w * 2 * D + D + c];
        }
        if (dgamma) atomicAdd(dgamma + c, g);
        if (dbeta) atomicAdd(dbeta + c, bsum);
    }
}

}  // namespace

extern "C" int edgedict_layernorm_fwd(int dtype, const void* x, const void* res, const float* gamma,
                                      const float* beta, void* y, float* mean, float* rstd, int B,
                                      int T, int D, int reduce, float eps, void* stream_) {
    ED_CHECK_ARG(dtype == ED_F32 || dtype == ED_BF16, "layernorm_fwd: bad dtype %d", dtype);
    ED_CHECK_ARG(reduce == 1 || reduce == 2, "layernorm_fwd: time reduction factor must be 1 or 2 (got %d)", reduce);
    ED_CHECK_ARG(B >= 0 && T >= 0 && D > 0, "layernorm_fwd: bad shape");
    if (B == 0 || T == 0) return ED_OK;
    ED_CHECK_ARG(x && gamma && beta && y && mean && rstd, "layernorm_fwd: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    const int vec = dtype == ED_F32 ? 4 : 8;
    const int vec_ok = (D % vec == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) &&
                       (!res || (uintptr_t)res % 16 == 0) && ((uintptr_t)gamma % 16 == 0) &&
                       ((uintptr_t)beta % 16 == 0);
    const long long orows = (long long)B * ((T + reduce - 1) / reduce);
    const int grid = ed_grid_for(orows, 4, 256 * 16);
    if (dtype == ED_F32)
        hipLaunchKernelGGL(layernorm_fwd<float>, dim3(grid), dim3(256), 0, stream, (const float*)x,
                           (const float*)res, gamma, beta, (float*)y, mean, rstd, B, T, D, reduce,
                           eps, vec_ok);
    else
        hipLaunchKernelGGL(layernorm_fwd<bf16_t>, dim3(grid), dim3(256), 0, stream,
                           (const bf16_t*)x, (const bf16_t*)res, gamma, beta, (bf16_t*)y, mean,
                           rstd, B, T, D, reduce, eps, vec_ok);
    ED_CHECK_LAUNCH("layernorm_fwd");
    return ED_OK;
}

extern "C" int edgedict_layernorm_bwd(int dtype, const void* dout, const void* x, const void* res,
                                      const float* gamma, const float* mean, const float* rstd,
                                      void* ds, float* dgamma, float* dbeta, int B, int T, int D,
                                      int reduce, void* stream_) {
    ED_CHECK_ARG(dtype == ED_F32 || dtype == ED_BF16, "layernorm_bwd: bad dtype %d", dtype);
    ED_CHECK_ARG(reduce == 1 || reduce == 2, "layernorm_bwd: time reduction factor must be 1 or 2");
    ED_CHECK_ARG(B >= 0 && T >= 0 && D > 0, "layernorm_bwd: bad shape");
    if (B == 0 || T == 0) return ED_OK;
    ED_CHECK_ARG(dout && x && gamma && mean && rstd && ds, "layernorm_bwd: null pointer");
    const size_t lds = (size_t)4 * 2 * D * sizeof(float);
    ED_CHECK_ARG(lds <= 64 * 1024, "layernorm_bwd: D = %d too large for the LDS accumulators", D);
    hipStream_t stream = (hipStream_t)stream_;
    const long long rows = (long long)B * T;
    const int grid = ed_grid_for(rows, 4 * 8, 1024);  // >= 8 rows per wave: amortise the atomics
    if (dtype == ED_F32)
        hipLaunchKernelGGL(layernorm_bwd<float>, dim3(grid), dim3(256), lds, stream,
                           (const float*)dout, (const float*)x, (const float*)res, gamma, mean,
                           rstd, (float*)ds, dgamma, dbeta, B, T, D, reduce);
    else
        hipLaunchKernelGGL(layernorm_bwd<bf16_t>, dim3(grid), dim3(256), lds, stream,
                           (const bf16_t*)dout, (const bf16_t*)x, (const bf16_t*)res, gamma, mean,
                           rstd, (bf16_t*)ds, dgamma, dbeta, B, T, D, reduce);
    ED_CHECK_LAUNCH("layernorm_bwd");
    return ED_OK;
}
