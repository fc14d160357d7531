// Convolutional waveform front-end (FrontEnd, rnnt/models.py:313-365) for gfx950:
//   conv1 (1 -> C0, kernel k0, stride s0, causal) -> [GELU -> GroupNorm(1, C_in) -> conv]* -> LayerNorm(C)
// "causal": nn.Conv1d pads (k-1) on both sides and the block drops the last (k-1) output frames
// (rnnt/models.py:314-318,336-337), so output frame t reads input frames [t*s - (k-1), t*s].
// Activations are kept channels-last [B, T, C]: a convolution is then an im2col gather (this file)
// followed by one MFMA GEMM (gemm.hip) against the weight viewed as [C_out, C_in*k]; the block's
// output [B, T, C] is what the final LayerNorm and the encoder consume (rnnt/models.py:361-364).
// Kernels: im2col / col2im (gather form, no atomics), exact-erf GELU forward/backward,
// GroupNorm with ONE group (statistics over all (t, c) of a sample, affine per channel) forward/backward.
// This is the compatibility front-end of cli/train.py, not the north-star log-mel path: kernels are
// plain bandwidth-shaped loops with 32-bit indexing, no fusion beyond what is written here.
#include "common.hpp"

namespace {

// cols[b, t, c*k + j] = x[b, t*s - p + j, c]   (0 outside [0, Tin))
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void fe_im2col(const TI* __restrict__ x, TO* __restrict__ cols, int B,
                                                 int Tin, int C, int Tout, int k, int s, int p) {
    const int CK = C * k;
    const long long n = (long long)B * Tout * CK;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int q = (int)(i % CK);
        const long long bt = i / CK;
        const int t = (int)(bt % Tout), b = (int)(bt / Tout);
        const int c = q / k, j = q - c * k;
        const int ti = t * s - p + j;
        float v = 0.f;
        if (ti >= 0 && ti < Tin) v = ElemIO<TI>::load(x + ((long long)b * Tin + ti) * C + c);
        ElemIO<TO>::store(cols + i, v);
    }
}

// dx[b, ti, c] = sum over (t, j) with t*s - p + j == ti of dcols[b, t, c*k + j]
template <typename T>
__global__ __launch_bounds__(256) void fe_col2im(const T* __restrict__ dcols, T* __restrict__ dx, int B,
                                                 int Tin, int C, int Tout, int k, int s, int p) {
    const long long n = (long long)B * Tin * C;
    const int CK = C * k;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long bt = i / C;
        const int ti = (int)(bt % Tin), b = (int)(bt / Tin);
        float acc = 0.f;
        for (int j = 0; j < k; ++j) {
            const int num = ti + p - j;
            if (num < 0 || num % s) continue;
            const int t = num / s;
            if (t < Tout) acc += ElemIO<T>::load(dcols + ((long long)b * Tout + t) * CK + c * k + j);
        }
        ElemIO<T>::store(dx + i, acc);
    }
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_df(float x) {
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
    return cdf + x * 0.39894228040143268f * expf(-0.5f * x * x);
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// per sample b: a = GELU(y); mean / rstd of a over all (t, c).  One workgroup per sample.
template <typename T>
__global__ __launch_bounds__(256) void fe_gelu_stats(const T* __restrict__ y, float* __restrict__ mean,
                                                     float* __restrict__ rstd, long long per, float eps) {
    __shared__ float sh[4];
    const int b = blockIdx.x;
    const T* yb = y + (long long)b * per;
    float s = 0.f;
    for (long long i = threadIdx.x; i < per; i += 256) s += gelu_f(ElemIO<T>::load(yb + i));
    const float m = block_sum(s, sh) / (float)per;
    float q = 0.f;
    for (long long i = threadIdx.x; i < per; i += 256) {
        const float d = gelu_f(ElemIO<T>::load(yb + i)) - m;
        q += d * d;
    }
    const float var = block_sum(q, sh) / (float)per;
    if (threadIdx.x == 0) {
        mean[b] = m;
        rstd[b] = rsqrtf(var + eps);
    }
}

// out[b,t,c] = (GELU(y) - mean_b) rstd_b gamma_c + beta_c
template <typename T>
__global__ __launch_bounds__(256) void fe_gelu_gn_apply(const T* __restrict__ y, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ out,
                                                        int B, long long per, int C) {
    const long long n = (long long)B * per;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / per), c = (int)(i % C);
        ElemIO<T>::store(out + i, (gelu_f(ElemIO<T>::load(y + i)) - mean[b]) * rstd[b] * gamma[c] + beta[c]);
    }
}

// backward, pass 1 (one workgroup per sample): s1 = mean(g), s2 = mean(g xhat), g = dout gamma_c
template <typename T>
__global__ __launch_bounds__(256) void fe_gn_bwd_stats(const T* __restrict__ y, const T* __restrict__ dout,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, float* __restrict__ s1,
                                                       float* __restrict__ s2, long long per, int C) {
    __shared__ float sh[4];
    const int b = blockIdx.x;
    const float m = mean[b], rs = rstd[b];
    float a1 = 0.f, a2 = 0.f;
    for (long long i = threadIdx.x; i < per; i += 256) {
        const long long o = (long long)b * per + i;
        const float g = ElemIO<T>::load(dout + o) * gamma[(int)(i % C)];
        const float xh = (gelu_f(ElemIO<T>::load(y + o)) - m) * rs;
        a1 += g;
        a2 += g * xh;
    }
    const float t1 = block_sum(a1, sh), t2 = block_sum(a2, sh);
    if (threadIdx.x == 0) {
        s1[b] = t1 / (float)per;
        s2[b] = t2 / (float)per;
    }
}

// pass 2: dy = GELU'(y) rstd (g - s1 - xhat s2); per-workgroup partial dgamma / dbeta rows
// (channel c = column; a workgroup owns a slab of (b, t) rows), summed by a second tiny kernel
template <typename T>
__global__ __launch_bounds__(256) void fe_gn_bwd_apply(const T* __restrict__ y, const T* __restrict__ dout,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ s1,
                                                       const float* __restrict__ s2, T* __restrict__ dy,
                                                       float* __restrict__ part, int B, int Tn, int C,
                                                       int rows_per_block) {
    // thread -> channel column c = threadIdx.x % C (C <= 256), row lane r = threadIdx.x / C
    const int rl = 256 / C, c = threadIdx.x % C, r0 = threadIdx.x / C;
    const long long rows = (long long)B * Tn;
    const long long rbeg = (long long)blockIdx.x * rows_per_block;
    const long long rend = rbeg + rows_per_block < rows ? rbeg + rows_per_block : rows;
    float dg = 0.f, db = 0.f;
    if (r0 < rl) {
        const float gm = gamma[c];
        for (long long r = rbeg + r0; r < rend; r += rl) {
            const int b = (int)(r / Tn);
            const long long o = r * C + c;
            const float yv = ElemIO<T>::load(y + o), d = ElemIO<T>::load(dout + o);
            const float xh = (gelu_f(yv) - mean[b]) * rstd[b];
            dg += d * xh;
            db += d;
            ElemIO<T>::store(dy + o, gelu_df(yv) * rstd[b] * (d * gm - s1[b] - xh * s2[b]));
        }
    }
    __shared__ float sg[256], sb[256];
    sg[threadIdx.x] = dg;
    sb[threadIdx.x] = db;
    __syncthreads();
    if (threadIdx.x < C) {
        float a = 0.f, bsum = 0.f;
        for (int q = 0; q < rl; ++q) { a += sg[q * C + threadIdx.x]; bsum += sb[q * C + threadIdx.x]; }
        part[((long long)blockIdx.x * 2) * C + threadIdx.x] = a;
        part[((long long)blockIdx.x * 2 + 1) * C + threadIdx.x] = bsum;
    }
}

__global__ void fe_sum_parts(const float* __restrict__ part, int nblocks, int C, float* __restrict__ dgamma,
                             float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float a = 0.f, b = 0.f;
    for (int i = 0; i < nblocks; ++i) {
        a += part[((long long)i * 2) * C + c];
        b += part[((long long)i * 2 + 1) * C + c];
    }
    dgamma[c] = a;
    dbeta[c] = b;
}

}  // namespace

extern "C" int edgedict_conv_out_frames(int Tin, int k, int s) {
    // nn.Conv1d(padding = k-1) then drop the last k-1 frames (rnnt/models.py:314-318,336-337)
    if (Tin <= 0 || k <= 0 || s <= 0) return 0;
    const int full = (Tin + 2 * (k - 1) - k) / s + 1;
    return full - (k - 1) > 0 ? full - (k - 1) : 0;
}

extern "C" int edgedict_conv_im2col(int in_dtype, int out_dtype, const void* x, void* cols, int B, int Tin,
                                    int C, int k, int s, void* stream) {
    ED_CHECK_ARG((in_dtype == ED_F32 || in_dtype == ED_BF16) && (out_dtype == ED_F32 || out_dtype == ED_BF16),
                 "conv_im2col: bad dtype");
    const int Tout = edgedict_conv_out_frames(Tin, k, s);
    ED_CHECK_ARG(B > 0 && C > 0 && Tout > 0, "conv_im2col: empty output (Tin=%d k=%d s=%d)", Tin, k, s);
    ED_CHECK_ARG(x && cols, "conv_im2col: null pointer");
    ED_CHECK_ARG(!(in_dtype == ED_BF16 && out_dtype == ED_F32), "conv_im2col: bf16 -> fp32 is not needed");
    hipStream_t st = (hipStream_t)stream;
    const int grid = ed_grid_for((long long)B * Tout * C * k, 256, 256 * 32);
    if (in_dtype == ED_F32 && out_dtype == ED_F32)
        hipLaunchKernelGGL((fe_im2col<float, float>), dim3(grid), dim3(256), 0, st, (const float*)x, (float*)cols, B, Tin, C, Tout, k, s, k - 1);
    else if (in_dtype == ED_F32)
        hipLaunchKernelGGL((fe_im2col<float, bf16_t>), dim3(grid), dim3(256), 0, st, (const float*)x, (bf16_t*)cols, B, Tin, C, Tout, k, s, k - 1);
    else
        hipLaunchKernelGGL((fe_im2col<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)cols, B, Tin, C, Tout, k, s, k - 1);
    ED_CHECK_LAUNCH("conv_im2col");
    return ED_OK;
}

extern "C" int edgedict_conv_col2im(int dtype, const void* dcols, void* dx, int B, int Tin, int C, int k,
                                    int s, void* stream) {
    ED_CHECK_ARG(dtype == ED_F32 || dtype == ED_BF16, "conv_col2im: bad dtype");
    const int Tout = edgedict_conv_out_frames(Tin, k, s);
    ED_CHECK_ARG(B > 0 && C > 0 && Tout > 0 && dcols && dx, "conv_col2im: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int grid = ed_grid_for((long long)B * Tin * C, 256, 256 * 32);
    if (dtype == ED_F32)
        hipLaunchKernelGGL(fe_col2im<float>, dim3(grid), dim3(256), 0, st, (const float*)dcols, (float*)dx, B, Tin, C, Tout, k, s, k - 1);
    else
        hipLaunchKernelGGL(fe_col2im<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)dcols, (bf16_t*)dx, B, Tin, C, Tout, k, s, k - 1);
    ED_CHECK_LAUNCH("conv_col2im");
    return ED_OK;
}

extern "C" int edgedict_gelu_groupnorm_fwd(int dtype, const void* y, const float* gamma, const float* beta,
                                           void* out, float* mean, float* rstd, int B, int T, int C,
                                           float eps, void* stream) {
    ED_CHECK_ARG(dtype == ED_F32 || dtype == ED_BF16, "gelu_groupnorm_fwd: bad dtype");
    ED_CHECK_ARG(B > 0 && T > 0 && C > 0 && y && gamma && beta && out && mean && rstd, "gelu_groupnorm_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const long long per = (long long)T * C;
    const int grid = ed_grid_for((long long)B * per, 256, 256 * 32);
    if (dtype == ED_F32) {
        hipLaunchKernelGGL(fe_gelu_stats<float>, dim3(B), dim3(256), 0, st, (const float*)y, mean, rstd, per, eps);
        hipLaunchKernelGGL(fe_gelu_gn_apply<float>, dim3(grid), dim3(256), 0, st, (const float*)y, mean, rstd, gamma, beta, (float*)out, B, per, C);
    } else {
        hipLaunchKernelGGL(fe_gelu_stats<bf16_t>, dim3(B), dim3(256), 0, st, (const bf16_t*)y, mean, rstd, per, eps);
        hipLaunchKernelGGL(fe_gelu_gn_apply<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)y, mean, rstd, gamma, beta, (bf16_t*)out, B, per, C);
    }
    ED_CHECK_LAUNCH("gelu_groupnorm_fwd");
    return ED_OK;
}

extern "C" size_t edgedict_gelu_groupnorm_bwd_workspace_bytes(int B, int T, int C) {
    if (B <= 0 || T <= 0 || C <= 0) return 0;
    const long long rows = (long long)B * T;
    const int rpb = 256;
    const long long nb = (rows + rpb - 1) / rpb;
    return (size_t)(2 * B + nb * 2 * C) * sizeof(float);
}

extern "C" int edgedict_gelu_groupnorm_bwd(int dtype, const void* y, const void* dout, const float* gamma,
                                           const float* mean, const float* rstd, void* dy, float* dgamma,
                                           float* dbeta, void* workspace, int B, int T, int C,
                                           void* stream) {
    ED_CHECK_ARG(dtype == ED_F32 || dtype == ED_BF16, "gelu_groupnorm_bwd: bad dtype");
    ED_CHECK_ARG(B > 0 && T > 0 && C > 0 && C <= 256, "gelu_groupnorm_bwd: need 0 < C <= 256 (C=%d)", C);
    ED_CHECK_ARG(y && dout && gamma && mean && rstd && dy && dgamma && dbeta && workspace, "gelu_groupnorm_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const long long per = (long long)T * C, rows = (long long)B * T;
    const int rpb = 256;
    const int nb = (int)((rows + rpb - 1) / rpb);
    float* s1 = (float*)workspace;
    float* s2 = s1 + B;
    float* part = s2 + B;
    if (dtype == ED_F32) {
        hipLaunchKernelGGL(fe_gn_bwd_stats<float>, dim3(B), dim3(256), 0, st, (const float*)y, (const float*)dout, mean, rstd, gamma, s1, s2, per, C);
        hipLaunchKernelGGL(fe_gn_bwd_apply<float>, dim3(nb), dim3(256), 0, st, (const float*)y, (const float*)dout, mean, rstd, gamma, s1, s2, (float*)dy, part, B, T, C, rpb);
    } else {
        hipLaunchKernelGGL(fe_gn_bwd_stats<bf16_t>, dim3(B), dim3(256), 0, st, (const bf16_t*)y, (const bf16_t*)dout, mean, rstd, gamma, s1, s2, per, C);
        hipLaunchKernelGGL(fe_gn_bwd_apply<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t*)y, (const bf16_t*)dout, mean, rstd, gamma, s1, s2, (bf16_t*)dy, part, B, T, C, rpb);
    }
    hipLaunchKernelGGL(fe_sum_parts, dim3((C + 63) / 64), dim3(64), 0, st, part, nb, C, dgamma, dbeta);
    ED_CHECK_LAUNCH("gelu_groupnorm_bwd");
    return ED_OK;
}
