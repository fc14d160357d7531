// Streaming (HBM-bound) helper kernels of the RNN-T path: dtype casts / transposes of weights,
// bias-gradient column sums, the prediction network's embedding lookup, the joint network's
// broadcast-add + tanh and its backward reduction, and the fused Adam update.
//
// Reference arithmetic:
//   Decoder.forward   rnnt/models.py:150-157  F.pad(ys,[1,0],BOS) -> nn.Embedding(padding_idx=PAD)
//   Joint.forward     rnnt/models.py:169-179  Linear(cat[enc;dec]) -> Tanh  (the first Linear is
//                     split as W1e*enc + W1d*dec + b1; here: tanh(E1[b,t]+D1[b,u]))
//   optimiser         cli/train.py:135-146,268 torch.optim.Adam (no weight decay)
#include <cstdlib>
#include "common.hpp"

namespace {

// ---------------------------------------------------------------- cast (contiguous)
template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ s, TD* __restrict__ d, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        ElemIO<TD>::store(d + i, ElemIO<TS>::load(s + i));
}

// ---------------------------------------------------------------- transpose + cast
// dst[c][r] = src[r][c];  32x32 tile through LDS, coalesced on both sides.
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void transpose_kernel(const TS* __restrict__ s,
                                                        TD* __restrict__ d, int R, int C) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + i * 8, c = c0 + tx;
        tile[ty + i * 8][tx] = (r < R && c < C) ? ElemIO<TS>::load(s + (long long)r * C + c) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + i * 8, r = r0 + tx;
        if (c < C && r < R) ElemIO<TD>::store(d + (long long)c * R + r, tile[tx][ty + i * 8]);
    }
}

// ---------------------------------------------------------------- column sum: out[n] += sum_m X[m][n]
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, long long ld,
                                                     float* __restrict__ out, long long M, int N,
                                                     int rows_per_block) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    const long long m0 = (long long)blockIdx.y * rows_per_block;
    const long long m1 = min(M, m0 + rows_per_block);
    if (col >= N) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    long long m = m0;
    for (; m + 3 < m1; m += 4) {
        a0 += ElemIO<T>::load(x + m * ld + col);
        a1 += ElemIO<T>::load(x + (m + 1) * ld + col);
        a2 += ElemIO<T>::load(x + (m + 2) * ld + col);
        a3 += ElemIO<T>::load(x + (m + 3) * ld + col);
    }
    for (; m < m1; ++m) a0 += ElemIO<T>::load(x + m * ld + col);
    atomicAdd(out + col, (a0 + a1) + (a2 + a3));
}

// ---------------------------------------------------------------- embedding
// out[b, u, :] = emb[tok(b,u), :], tok(b,0) = bos when prepend_bos else labels[b,u]
template <typename T, typename TW>
__global__ void embedding_fwd(const int32_t* __restrict__ tokens, int tok_stride,
                              const TW* __restrict__ emb, T* __restrict__ out, int B, int Uout,
                              int E, int prepend_bos, int bos, int V) {
    const long long n = (long long)B * Uout * E;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i % E);
        const long long bu = i / E;
        const int u = (int)(bu % Uout), b = (int)(bu / Uout);
        int tok;
        if (prepend_bos) tok = (u == 0) ? bos : tokens[(long long)b * tok_stride + u - 1];
        else tok = tokens[(long long)b * tok_stride + u];
        tok = min(max(tok, 0), V - 1);
        ElemIO<T>::store(out + i, ElemIO<TW>::load(emb + (long long)tok * E + e));
    }
}
// demb[tok] += dout rows (fp32 atomics); the padding row receives nothing (padding_idx)
template <typename T>
__global__ void embedding_bwd(const int32_t* __restrict__ tokens, int tok_stride,
                              const T* __restrict__ dout, float* __restrict__ demb, int B, int Uout,
                              int E, int prepend_bos, int bos, int pad, int V) {
    const long long n = (long long)B * Uout * E;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i % E);
        const long long bu = i / E;
        const int u = (int)(bu % Uout), b = (int)(bu / Uout);
        int tok;
        if (prepend_bos) tok = (u == 0) ? bos : tokens[(long long)b * tok_stride + u - 1];
        else tok = tokens[(long long)b * tok_stride + u];
        if (tok == pad || tok < 0 || tok >= V) continue;
        atomicAdd(demb + (long long)tok * E + e, ElemIO<T>::load(dout + i));
    }
}

// ---------------------------------------------------------------- dropout
// y = x * keep(i) / (1 - p), keep(i) = hash(seed, i) >= p: a counter-based mask (no state, no mask
// tensor): the backward pass regenerates it from the same (seed, index), so dy -> dx is the SAME
// kernel.  nn.Dropout / nn.LSTM(dropout=p) semantics (rnnt/models.py:47-53,145-147): scaling by
// 1/(1-p) in training, identity in eval (the host simply does not call it).
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, long long n, float p,
                               unsigned seed) {
    const unsigned thresh = ed_drop_thresh(p);              // keep iff hash >= p * 2^32 (common.hpp)
    const float scale = 1.f / (1.f - p);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        ElemIO<T>::store(y + i, ed_drop_keep(seed, i, thresh) ? ElemIO<T>::load(x + i) * scale : 0.f);
    }
}

// ---------------------------------------------------------------- SpecAugment masks
// x[b, t, f] = 0 where t or f falls in one of the row's mask intervals (TimeMasking /
// FrequencyMasking, rnnt/transforms.py:54-146: the reference applies them AFTER frame stacking, on
// [B, F, T] with F = n_mels*stack, the frequency interval running over the stacked feature index).
// The intervals are drawn on the host with Python's `random` in the reference's call order (so the
// same seed gives the same masks) and applied here in one pass over the resident batch.
// layout here: x [B, T, F] (the model's input layout)
__global__ void spec_mask_kernel(float* __restrict__ x, int B, int Tn, int F,
                                 const int32_t* __restrict__ t_iv, int n_t,   // [B][n_t][2] start,end
                                 const int32_t* __restrict__ f_iv, int n_f) { // [B][n_f][2]
    const long long n = (long long)B * Tn * F;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int f = (int)(i % F);
        const long long bt = i / F;
        const int t = (int)(bt % Tn), b = (int)(bt / Tn);
        bool kill = false;
        for (int k = 0; k < n_t; ++k) {
            const int s0 = t_iv[((long long)b * n_t + k) * 2], s1 = t_iv[((long long)b * n_t + k) * 2 + 1];
            kill |= (t >= s0 && t < s1);
        }
        for (int k = 0; k < n_f; ++k) {
            const int s0 = f_iv[((long long)b * n_f + k) * 2], s1 = f_iv[((long long)b * n_f + k) * 2 + 1];
            kill |= (f >= s0 && f < s1);
        }
        if (kill) x[i] = 0.f;
    }
}

// ---------------------------------------------------------------- joint: hid = tanh(E1[b,t] + D1[b,u])
template <typename T> __device__ __forceinline__ float joint_tanh(float x) { return tanhf(x); }
// bf16 mode: v_exp / v_rcp form (abs. error ~1e-7, the result is rounded to bf16 anyway)
template <> __device__ __forceinline__ float joint_tanh<bf16_t>(float x) {
    const float xc = fminf(fmaxf(x, -15.f), 15.f);
    return 1.f - __fdividef(2.f, 1.f + __expf(2.f * xc));
}

// One workgroup per encoder frame (b, t): a thread keeps its 16-byte column chunk of E1[b,t] in
// registers and walks u (32-bit index arithmetic only; the flat-index form spent more time in
// 64-bit divisions than in tanh).
template <typename T>
__global__ __launch_bounds__(256) void joint_hidden_fwd(const T* __restrict__ E1,
                                                        const T* __restrict__ D1,
                                                        T* __restrict__ hid, int B, int Tn, int U1,
                                                        int J, const int32_t* __restrict__ act_lens,
                                                        const int32_t* __restrict__ label_lens,
                                                        const long long* __restrict__ pk_off) {
    constexpr int VEC = ElemIO<T>::VEC;
    const int chunks = J / VEC;  // J % VEC == 0 checked on the host
    const int ulanes = max(1, 256 / chunks);          // u values handled side by side
    const int c = threadIdx.x % chunks, ul = threadIdx.x / chunks;
    if (ul >= ulanes) return;
    for (int bt = blockIdx.x; bt < B * Tn; bt += gridDim.x) {
        const int b = bt / Tn, t = bt - b * Tn;
        int Ub = U1 - 1;
        long long row0 = (long long)bt * U1;       // dense: row = (b*T + t)*U1 + u
        if (pk_off) {   // packed lattice: only cells inside the utterance's (T_b, U_b + 1) box exist
            Ub = label_lens[b];
            if (t >= act_lens[b]) continue;
            row0 = pk_off[b] + (long long)t * (Ub + 1);
        }
        float e[VEC];
        ElemIO<T>::load_vec(E1 + (long long)bt * J + c * VEC, e);
        const T* drow = D1 + (long long)b * U1 * J + c * VEC;
        T* orow = hid + row0 * J + c * VEC;
        for (int u = ul; u <= Ub; u += ulanes) {
            float d[VEC], o[VEC];
            ElemIO<T>::load_vec(drow + (long long)u * J, d);
#pragma unroll
            for (int k = 0; k < VEC; ++k) o[k] = joint_tanh<T>(e[k] + d[k]);
            ElemIO<T>::store_vec(orow + (long long)u * J, o);
        }
    }
}

// backward of the broadcast-add + tanh:  dpre = dhid * (1 - hid^2)
//   dE1[b,t,j] = sum_u dpre[b,t,u,j]        dD1[b,u,j] = sum_t dpre[b,t,u,j]
// HBM-bound (reads 2 x B*T*U1*J elements once).  One workgroup per (64-wide j block, b, t slab);
// a wave is 8 j-vectors (8 elements = 16 B bf16 / 32 B f32 per lane, so one (b,t,u) row segment
// is a full 128-byte line) x 8 label lanes; lane (jv, ug) owns label positions u = ug + 8 i.
// The four waves take alternating frames.  Sums over u: registers + three xor-shuffles; sums
// over t: per-thread registers (JB_NU x 8 accumulators), combined across the waves through LDS
// once per slab and across slabs with fp32 atomics.
constexpr int JB_NU = 9;              // label positions per lane and pass (8 * 9 = 72 >= U+1 = 65)
constexpr int JB_UCHUNK = 8 * JB_NU;
template <typename T> struct Load8;
template <> struct Load8<bf16_t> {
    __device__ static __forceinline__ void ld(const bf16_t* p, float (&o)[8]) {
        ElemIO<bf16_t>::load_vec(p, o);
    }
};
template <> struct Load8<float> {
    __device__ static __forceinline__ void ld(const float* p, float (&o)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(p);
        const float4 b = *reinterpret_cast<const float4*>(p + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
        o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    }
};
template <typename T>
__global__ __launch_bounds__(256) void joint_hidden_bwd(const T* __restrict__ dhid,
                                                        const T* __restrict__ hid,
                                                        float* __restrict__ dE1,
                                                        float* __restrict__ dD1, int B, int Tn,
                                                        int U1, int J, int t_per_block,
                                                        const int32_t* __restrict__ act_lens,
                                                        const int32_t* __restrict__ label_lens,
                                                        const long long* __restrict__ pk_off) {
    __shared__ float4 red[2][JB_NU][2][64];   // lane-wise hand-off of partial label sums (36 KB)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int jv = lane & 7, ug = lane >> 3;
    const int j0 = blockIdx.x * 64 + jv * 8;
    const int b = blockIdx.y;
    const int t0 = blockIdx.z * t_per_block, t1 = min(Tn, t0 + t_per_block);
    const bool jlive = j0 < J;   // J % 8 == 0 checked on the host
    // packed lattice: utterance b owns rows pk_off[b] + t*(U_b+1) + u, t < T_b, u <= U_b
    const int Tb = pk_off ? act_lens[b] : Tn;
    const int Ulim = pk_off ? label_lens[b] + 1 : U1;
    const long long rbase = pk_off ? pk_off[b] : (long long)b * Tn * U1;
    for (int uc = 0; uc < U1; uc += JB_UCHUNK) {
        float usum[JB_NU][8];
#pragma unroll
        for (int i = 0; i < JB_NU; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) usum[i][e] = 0.f;
        for (int t = t0 + wave; t < t1; t += 4) {
            const long long base = (rbase + (long long)t * Ulim) * J + j0;
            float tsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < JB_NU; ++i) {
                const int u = uc + ug + 8 * i;
                if (u < Ulim && t < Tb && jlive) {
                    float h[8], g[8];
                    Load8<T>::ld(hid + base + (long long)u * J, h);
                    Load8<T>::ld(dhid + base + (long long)u * J, g);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float dp = g[e] * (1.f - h[e] * h[e]);
                        tsum[e] += dp;
                        usum[i][e] += dp;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                tsum[e] += __shfl_xor(tsum[e], 8, 64);
                tsum[e] += __shfl_xor(tsum[e], 16, 64);
                tsum[e] += __shfl_xor(tsum[e], 32, 64);
            }
            if (ug == 0 && jlive) {
                float* de = dE1 + ((long long)b * Tn + t) * J + j0;
                if (uc == 0) {
                    *reinterpret_cast<float4*>(de) = make_float4(tsum[0], tsum[1], tsum[2], tsum[3]);
                    *reinterpret_cast<float4*>(de + 4) = make_float4(tsum[4], tsum[5], tsum[6], tsum[7]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) de[e] += tsum[e];
                }
            }
        }
        // combine the four waves' label sums lane-wise (lane l of every wave owns the same (u, j)
        // set): waves 2,3 -> waves 0,1, then wave 1 -> wave 0; one atomic per (u, j) and slab
        if (uc > 0) __syncthreads();
        if (wave >= 2) {
#pragma unroll
            for (int i = 0; i < JB_NU; ++i) {
                red[wave - 2][i][0][lane] = make_float4(usum[i][0], usum[i][1], usum[i][2], usum[i][3]);
                red[wave - 2][i][1][lane] = make_float4(usum[i][4], usum[i][5], usum[i][6], usum[i][7]);
            }
        }
        __syncthreads();
        if (wave < 2) {
#pragma unroll
            for (int i = 0; i < JB_NU; ++i) {
                const float4 p = red[wave][i][0][lane], q = red[wave][i][1][lane];
                usum[i][0] += p.x; usum[i][1] += p.y; usum[i][2] += p.z; usum[i][3] += p.w;
                usum[i][4] += q.x; usum[i][5] += q.y; usum[i][6] += q.z; usum[i][7] += q.w;
            }
        }
        __syncthreads();
        if (wave == 1) {
#pragma unroll
            for (int i = 0; i < JB_NU; ++i) {
                red[0][i][0][lane] = make_float4(usum[i][0], usum[i][1], usum[i][2], usum[i][3]);
                red[0][i][1][lane] = make_float4(usum[i][4], usum[i][5], usum[i][6], usum[i][7]);
            }
        }
        __syncthreads();
        if (wave == 0 && jlive) {
#pragma unroll
            for (int i = 0; i < JB_NU; ++i) {
                const int u = uc + ug + 8 * i;
                if (u >= U1) continue;
                const float4 p = red[0][i][0][lane], q = red[0][i][1][lane];
                float* dd = dD1 + ((long long)b * U1 + u) * J + j0;
                atomicAdd(dd + 0, usum[i][0] + p.x); atomicAdd(dd + 1, usum[i][1] + p.y);
                atomicAdd(dd + 2, usum[i][2] + p.z); atomicAdd(dd + 3, usum[i][3] + p.w);
                atomicAdd(dd + 4, usum[i][4] + q.x); atomicAdd(dd + 5, usum[i][5] + q.y);
                atomicAdd(dd + 6, usum[i][6] + q.z); atomicAdd(dd + 7, usum[i][7] + q.w);
            }
        }
    }
}

// ---------------------------------------------------------------- Adam (torch.optim.Adam semantics)
// m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= step_size * m / (sqrt(v)/sqrt(bc2) + eps)
// grad_scale is read from device memory (gradient clipping coefficient) when given.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                            float* __restrict__ m, float* __restrict__ v, long long n, float lr,
                            float b1, float b2, float omb1, float omb2, float eps, float bc1, float bc2,
                            float weight_decay, float grad_scale_host,
                            const float* __restrict__ grad_scale,
                            bf16_t* __restrict__ p_bf16, const unsigned* __restrict__ skip) {
    // guard (edgedict_adam_step_guarded): a bounded in-kernel wait of this step's encoder stack gave up, so
    // its gradients are garbage - leave p, m, v untouched; the host raises at its next check of the word.
    // `skip` is a DEVICE word filled by guard_fetch_kernel just before this launch (the give-up words live in
    // pinned host memory: reading them from every workgroup here cost 0.3-1 ms per step over PCIe).
    if (skip && *skip != 0u) return;
    const float gs = grad_scale_host * (grad_scale ? *grad_scale : 1.f);
    const float step = lr / bc1;
    const float inv_sqrt_bc2 = rsqrtf(bc2);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        float gi = g[i] * gs;
        float pi = p[i];
        if (weight_decay != 0.f) gi += weight_decay * pi;
        const float mi = b1 * m[i] + omb1 * gi;
        const float vi = b2 * v[i] + omb2 * gi * gi;
        m[i] = mi;
        v[i] = vi;
        pi -= step * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
        p[i] = pi;
        if (p_bf16) p_bf16[i] = f32_to_bf16(pi);
    }
}

// scratch[0] = OR of the n device-visible (host-pinned) give-up words: one lane, n PCIe reads
__global__ void guard_fetch_kernel(const unsigned* words, int n, unsigned* scratch) {
    if (threadIdx.x == 0) {
        unsigned any = 0u;
        for (int k = 0; k < n; ++k) any |= __hip_atomic_load(words + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        *scratch = any;
    }
}

// sum of squares -> out[0] (atomic); used for the global gradient norm
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long long n,
                                                    float* __restrict__ out) {
    __shared__ float part[4];
    float a = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        a += x[i] * x[i];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}
// coef[0] = min(1, max_norm / (sqrt(sumsq) + 1e-6))   (torch.nn.utils.clip_grad_norm_)
__global__ void clip_coef_kernel(const float* sumsq, float max_norm, float pre_scale, float* coef,
                                 float* norm_out) {
    const float nrm = sqrtf(*sumsq) * pre_scale;
    if (norm_out) *norm_out = nrm;
    *coef = fminf(1.f, max_norm / (nrm + 1e-6f));
}

}  // namespace

extern "C" int edgedict_cast(int src_dtype, const void* src, int dst_dtype, void* dst,
                             long long n, void* stream_) {
    ED_CHECK_ARG(n >= 0, "cast: negative size");
    if (n == 0) return ED_OK;
    ED_CHECK_ARG(src && dst, "cast: null pointer");
    hipStream_t s = (hipStream_t)stream_;
    const int grid = ed_grid_for(n, 256 * 4, 256 * 8);
    if (src_dtype == ED_F32 && dst_dtype == ED_BF16)
        hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(grid), dim3(256), 0, s, (const float*)src, (bf16_t*)dst, n);
    else if (src_dtype == ED_BF16 && dst_dtype == ED_F32)
        hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(grid), dim3(256), 0, s, (const bf16_t*)src, (float*)dst, n);
    else if (src_dtype == ED_F32 && dst_dtype == ED_F32)
        hipLaunchKernelGGL((cast_kernel<float, float>), dim3(grid), dim3(256), 0, s, (const float*)src, (float*)dst, n);
    else if (src_dtype == ED_BF16 && dst_dtype == ED_BF16)
        hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, n);
    else {
        ed_set_error("cast: unsupported dtype pair %d -> %d", src_dtype, dst_dtype);
        return ED_ERR_INVALID;
    }
    ED_CHECK_LAUNCH("cast");
    return ED_OK;
}

extern "C" int edgedict_transpose(int src_dtype, const void* src, int dst_dtype, void* dst, int R,
                                  int C, void* stream_) {
    ED_CHECK_ARG(R >= 0 && C >= 0, "transpose: negative size");
    if (R == 0 || C == 0) return ED_OK;
    ED_CHECK_ARG(src && dst, "transpose: null pointer");
    hipStream_t s = (hipStream_t)stream_;
    dim3 grid((C + 31) / 32, (R + 31) / 32);
    if (src_dtype == ED_F32 && dst_dtype == ED_BF16)
        hipLaunchKernelGGL((transpose_kernel<float, bf16_t>), grid, dim3(256), 0, s, (const float*)src, (bf16_t*)dst, R, C);
    else if (src_dtype == ED_F32 && dst_dtype == ED_F32)
        hipLaunchKernelGGL((transpose_kernel<float, float>), grid, dim3(256), 0, s, (const float*)src, (float*)dst, R, C);
    else if (src_dtype == ED_BF16 && dst_dtype == ED_BF16)
        hipLaunchKernelGGL((transpose_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, R, C);
    else if (src_dtype == ED_BF16 && dst_dtype == ED_F32)
        hipLaunchKernelGGL((transpose_kernel<bf16_t, float>), grid, dim3(256), 0, s, (const bf16_t*)src, (float*)dst, R, C);
    else {
        ed_set_error("transpose: unsupported dtype pair %d -> %d", src_dtype, dst_dtype);
        return ED_ERR_INVALID;
    }
    ED_CHECK_LAUNCH("transpose");
    return ED_OK;
}

extern "C" int edgedict_colsum(int dtype, const void* x, long long ld, float* out, long long M,
                               int N, void* stream_) {
    ED_CHECK_ARG(dtype == ED_F32 || dtype == ED_BF16, "colsum: bad dtype");
    ED_CHECK_ARG(M >= 0 && N >= 0, "colsum: negative size");
    if (M == 0 || N == 0) return ED_OK;
    ED_CHECK_ARG(x && out, "colsum: null pointer");
    hipStream_t s = (hipStream_t)stream_;
    // (a 16-byte-per-thread variant of this kernel was measured in round 6: beside the encoder's BPTT it cost the step
    // 0.45 ms - profiles/r6_colsum.txt - and was removed; the one big column sum of the step is fused into rnnt_grad)
    const int colblocks = (N + 255) / 256;
    long long want_rowblocks = (2048 + colblocks - 1) / colblocks;  // ~2k workgroups in total
    long long rpb = (M + want_rowblocks - 1) / want_rowblocks;
    if (rpb < 64) rpb = 64;
    const long long rowblocks = (M + rpb - 1) / rpb;
    ED_CHECK_ARG(rowblocks <= 65535, "colsum: too many row blocks");
    dim3 grid(colblocks, (unsigned)rowblocks);
    if (dtype == ED_F32)
        hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)x, ld, out, M, N, (int)rpb);
    else
        hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, ld, out, M, N, (int)rpb);
    ED_CHECK_LAUNCH("colsum");
    return ED_OK;
}

extern "C" int edgedict_embedding_fwd(int out_dtype, int emb_dtype, const int32_t* tokens,
                                      int tok_stride, const void* emb, void* out, int B, int Uout,
                                      int E, int V, int prepend_bos, int bos, void* stream_) {
    ED_CHECK_ARG(B >= 0 && Uout >= 0 && E > 0 && V > 0, "embedding_fwd: bad shape");
    if (B == 0 || Uout == 0) return ED_OK;
    ED_CHECK_ARG(emb && out && (tokens || (prepend_bos && Uout == 1)), "embedding_fwd: null pointer");
    hipStream_t s = (hipStream_t)stream_;
    const long long n = (long long)B * Uout * E;
    const int grid = ed_grid_for(n, 256, 2048);
#define ED_EMB(TO, TW) hipLaunchKernelGGL((embedding_fwd<TO, TW>), dim3(grid), dim3(256), 0, s, tokens, tok_stride, (const TW*)emb, (TO*)out, B, Uout, E, prepend_bos, bos, V)
    if (out_dtype == ED_F32 && emb_dtype == ED_F32) ED_EMB(float, float);
    else if (out_dtype == ED_BF16 && emb_dtype == ED_F32) ED_EMB(bf16_t, float);
    else if (out_dtype == ED_BF16 && emb_dtype == ED_BF16) ED_EMB(bf16_t, bf16_t);
    else {
        ed_set_error("embedding_fwd: unsupported dtype pair");
        return ED_ERR_INVALID;
    }
#undef ED_EMB
    ED_CHECK_LAUNCH("embedding_fwd");
    return ED_OK;
}

extern "C" int edgedict_embedding_bwd(int dtype, const int32_t* tokens, int tok_stride,
                                      const void* dout, float* demb, int B, int Uout, int E, int V,
                                      int prepend_bos, int bos, int pad, void* stream_) {
    ED_CHECK_ARG(B >= 0 && Uout >= 0 && E > 0 && V > 0, "embedding_bwd: bad shape");
    if (B == 0 || Uout == 0) return ED_OK;
    ED_CHECK_ARG(dout && demb && (tokens || (prepend_bos && Uout == 1)), "embedding_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream_;
    const long long n = (long long)B * Uout * E;
    const int grid = ed_grid_for(n, 256, 2048);
    if (dtype == ED_F32)
        hipLaunchKernelGGL(embedding_bwd<float>, dim3(grid), dim3(256), 0, s, tokens, tok_stride, (const float*)dout, demb, B, Uout, E, prepend_bos, bos, pad, V);
    else if (dtype == ED_BF16)
        hipLaunchKernelGGL(embedding_bwd<bf16_t>, dim3(grid), dim3(256), 0, s, tokens, tok_stride, (const bf16_t*)dout, demb, B, Uout, E, prepend_bos, bos, pad, V);
    else {
        ed_set_error("embedding_bwd: bad dtype");
        return ED_ERR_INVALID;
    }
    ED_CHECK_LAUNCH("embedding_bwd");
    return ED_OK;
}

extern "C" int edgedict_dropout(int dtype, const void* x, void* y, long long n, float p,
                                unsigned seed, void* stream_) {
    ED_CHECK_ARG(dtype == ED_F32 || dtype == ED_BF16, "dropout: bad dtype");
    ED_CHECK_ARG(p >= 0.f && p < 1.f, "dropout: p must be in [0, 1)");
    ED_CHECK_ARG(n >= 0, "dropout: bad size");
    if (n == 0) return ED_OK;
    ED_CHECK_ARG(x && y, "dropout: null pointer");
    const int grid = ed_grid_for(n, 256 * 4, 256 * 16);
    if (dtype == ED_F32)
        hipLaunchKernelGGL(dropout_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream_, (const float*)x, (float*)y, n, p, seed);
    else
        hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream_, (const bf16_t*)x, (bf16_t*)y, n, p, seed);
    ED_CHECK_LAUNCH("dropout");
    return ED_OK;
}

extern "C" int edgedict_spec_mask(float* x, int B, int T, int F, const int32_t* t_iv, int n_t,
                                  const int32_t* f_iv, int n_f, void* stream_) {
    ED_CHECK_ARG(B >= 0 && T >= 0 && F > 0, "spec_mask: bad shape");
    ED_CHECK_ARG(n_t >= 0 && n_f >= 0 && (n_t == 0 || t_iv) && (n_f == 0 || f_iv), "spec_mask: bad intervals");
    if (B == 0 || T == 0 || (n_t == 0 && n_f == 0)) return ED_OK;
    ED_CHECK_ARG(x, "spec_mask: null pointer");
    hipLaunchKernelGGL(spec_mask_kernel, dim3(ed_grid_for((long long)B * T * F, 256 * 4, 256 * 16)),
                       dim3(256), 0, (hipStream_t)stream_, x, B, T, F, t_iv, n_t, f_iv, n_f);
    ED_CHECK_LAUNCH("spec_mask");
    return ED_OK;
}

static int joint_hidden_fwd_impl(int dtype, const void* E1, const void* D1, void* hid, int B, int T,
                                 int U1, int J, const int32_t* act_lens, const int32_t* label_lens,
                                 const long long* pk_off, void* stream_) {
    ED_CHECK_ARG(dtype == ED_F32 || dtype == ED_BF16, "joint_hidden_fwd: bad dtype");
    ED_CHECK_ARG(B > 0 && T > 0 && U1 > 0 && J > 0, "joint_hidden_fwd: bad shape");
    const int vec = dtype == ED_F32 ? 4 : 8;
    ED_CHECK_ARG(J % vec == 0, "joint_hidden_fwd: joint size %d must be a multiple of %d", J, vec);
    ED_CHECK_ARG(E1 && D1 && hid, "joint_hidden_fwd: null pointer");
    hipStream_t s = (hipStream_t)stream_;
    ED_CHECK_ARG(J / vec <= 256, "joint_hidden_fwd: joint size %d too large", J);
    const int grid = ed_grid_for((long long)B * T, 1, 256 * 64);
    if (dtype == ED_F32)
        hipLaunchKernelGGL(joint_hidden_fwd<float>, dim3(grid), dim3(256), 0, s, (const float*)E1, (const float*)D1, (float*)hid, B, T, U1, J, act_lens, label_lens, pk_off);
    else
        hipLaunchKernelGGL(joint_hidden_fwd<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)E1, (const bf16_t*)D1, (bf16_t*)hid, B, T, U1, J, act_lens, label_lens, pk_off);
    ED_CHECK_LAUNCH("joint_hidden_fwd");
    return ED_OK;
}

extern "C" int edgedict_joint_hidden_fwd(int dtype, const void* E1, const void* D1, void* hid,
                                         int B, int T, int U1, int J, void* stream_) {
    return joint_hidden_fwd_impl(dtype, E1, D1, hid, B, T, U1, J, nullptr, nullptr, nullptr, stream_);
}

extern "C" int edgedict_joint_hidden_fwd_packed(int dtype, const void* E1, const void* D1, void* hid,
                                                const int32_t* act_lens, const int32_t* label_lens,
                                                const long long* row_offsets, int B, int T, int U1,
                                                int J, void* stream_) {
    ED_CHECK_ARG(act_lens && label_lens && row_offsets, "joint_hidden_fwd_packed: null lengths/offsets");
    return joint_hidden_fwd_impl(dtype, E1, D1, hid, B, T, U1, J, act_lens, label_lens, row_offsets, stream_);
}

static int joint_hidden_bwd_impl(int dtype, const void* dhid, const void* hid, float* dE1, float* dD1,
                                 int B, int T, int U1, int J, const int32_t* act_lens,
                                 const int32_t* label_lens, const long long* pk_off, void* stream_) {
    ED_CHECK_ARG(dtype == ED_F32 || dtype == ED_BF16, "joint_hidden_bwd: bad dtype");
    ED_CHECK_ARG(B > 0 && T > 0 && U1 > 0 && J > 0, "joint_hidden_bwd: bad shape");
    ED_CHECK_ARG(dhid && hid && dE1 && dD1, "joint_hidden_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream_;
    // dD1 is accumulated with atomics across t slabs: zero it first
    hipError_t e = hipMemsetAsync(dD1, 0, (size_t)B * U1 * J * sizeof(float), s);
    if (e != hipSuccess) {
        ed_set_error("joint_hidden_bwd: memset failed: %s", hipGetErrorString(e));
        return ED_ERR_LAUNCH;
    }
    ED_CHECK_ARG(J % 8 == 0, "joint_hidden_bwd: joint size %d must be a multiple of 8", J);
    const int jblocks = (J + 63) / 64;
    static const int wg_target = [] { const char* e = getenv("EDGEDICT_JHB_WGS"); return e && atoi(e) > 0 ? atoi(e) : 1280; }();
    // t slabs per (utterance, 64-column block): every slab ends with U1 x 64 atomics into dD1, so more workgroups are
    // not better - E6D2 bench batch: 640 / 1280 / 2048 / 4096 / 8192 workgroups = 0.52 / 0.45 / 0.49 / 0.63 / 1.0 ms
    int tslabs = (wg_target + B * jblocks - 1) / (B * jblocks);
    if (tslabs > (T + 7) / 8) tslabs = (T + 7) / 8;         // at least two frames per wave
    if (tslabs < 1) tslabs = 1;
    const int tpb = (T + tslabs - 1) / tslabs;
    tslabs = (T + tpb - 1) / tpb;
    dim3 grid(jblocks, B, tslabs);
    if (dtype == ED_F32)
        hipLaunchKernelGGL(joint_hidden_bwd<float>, grid, dim3(256), 0, s, (const float*)dhid, (const float*)hid, dE1, dD1, B, T, U1, J, tpb, act_lens, label_lens, pk_off);
    else
        hipLaunchKernelGGL(joint_hidden_bwd<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)dhid, (const bf16_t*)hid, dE1, dD1, B, T, U1, J, tpb, act_lens, label_lens, pk_off);
    ED_CHECK_LAUNCH("joint_hidden_bwd");
    return ED_OK;
}

extern "C" int edgedict_joint_hidden_bwd(int dtype, const void* dhid, const void* hid, float* dE1,
                                         float* dD1, int B, int T, int U1, int J, void* stream_) {
    return joint_hidden_bwd_impl(dtype, dhid, hid, dE1, dD1, B, T, U1, J, nullptr, nullptr, nullptr, stream_);
}

extern "C" int edgedict_joint_hidden_bwd_packed(int dtype, const void* dhid, const void* hid,
                                                float* dE1, float* dD1, const int32_t* act_lens,
                                                const int32_t* label_lens,
                                                const long long* row_offsets, int B, int T, int U1,
                                                int J, void* stream_) {
    ED_CHECK_ARG(act_lens && label_lens && row_offsets, "joint_hidden_bwd_packed: null lengths/offsets");
    return joint_hidden_bwd_impl(dtype, dhid, hid, dE1, dD1, B, T, U1, J, act_lens, label_lens, row_offsets, stream_);
}

extern "C" int edgedict_adam_step_guarded(float* p, const float* g, float* m, float* v, long long n,
                                          float lr, float beta1, float beta2, float eps, int step,
                                          float weight_decay, float grad_scale_host,
                                          const float* grad_scale, void* p_bf16,
                                          const unsigned* skip_words, int n_skip, unsigned* guard_scratch,
                                          void* stream_) {
    ED_CHECK_ARG(n >= 0 && step >= 1, "adam_step: bad size/step");
    if (n == 0) return ED_OK;
    ED_CHECK_ARG(p && g && m && v, "adam_step: null pointer");
    ED_CHECK_ARG(n_skip >= 0 && n_skip <= 16 && (n_skip == 0 || (skip_words && guard_scratch)),
                 "adam_step_guarded: bad guard words / scratch");
    if (n_skip > 0) {
        hipLaunchKernelGGL(guard_fetch_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, skip_words, n_skip, guard_scratch);
        ED_CHECK_LAUNCH("guard_fetch_kernel");
    }
    // torch.optim.Adam works out 1 - beta and 1 - beta^step in DOUBLE from the Python floats and only
    // then rounds to fp32; the betas arrive here as fp32 images of decimal literals, and 1.f - 0.999f
    // differs from (float)(1 - 0.999) by 1.3e-5 relative (it shows in exp_avg_sq).  Recover the decimal
    // (7 significant digits) and do the same arithmetic in double.
    const double b1d = nearbyint((double)beta1 * 1e7) / 1e7, b2d = nearbyint((double)beta2 * 1e7) / 1e7;
    const float omb1 = (float)(1.0 - b1d), omb2 = (float)(1.0 - b2d);
    const float bc1 = (float)(1.0 - pow(b1d, (double)step));
    const float bc2 = (float)(1.0 - pow(b2d, (double)step));
    hipLaunchKernelGGL(adam_kernel, dim3(ed_grid_for(n, 256 * 4, 256 * 8)), dim3(256), 0,
                       (hipStream_t)stream_, p, g, m, v, n, lr, beta1, beta2, omb1, omb2, eps, bc1, bc2,
                       weight_decay, grad_scale_host, grad_scale, (bf16_t*)p_bf16, n_skip > 0 ? guard_scratch : nullptr);
    ED_CHECK_LAUNCH("adam_step");
    return ED_OK;
}

extern "C" int edgedict_adam_step(float* p, const float* g, float* m, float* v, long long n,
                                  float lr, float beta1, float beta2, float eps, int step,
                                  float weight_decay, float grad_scale_host,
                                  const float* grad_scale, void* p_bf16, void* stream_) {
    return edgedict_adam_step_guarded(p, g, m, v, n, lr, beta1, beta2, eps, step, weight_decay, grad_scale_host,
                                      grad_scale, p_bf16, nullptr, 0, nullptr, stream_);
}

extern "C" int edgedict_grad_clip_coef(const float* g, long long n, float max_norm,
                                       float pre_scale, float* sumsq_ws, float* coef,
                                       float* norm_out, void* stream_) {
    ED_CHECK_ARG(n >= 0 && g && sumsq_ws && coef, "grad_clip_coef: bad arguments");
    hipStream_t s = (hipStream_t)stream_;
    hipError_t e = hipMemsetAsync(sumsq_ws, 0, sizeof(float), s);
    if (e != hipSuccess) {
        ed_set_error("grad_clip_coef: memset failed: %s", hipGetErrorString(e));
        return ED_ERR_LAUNCH;
    }
    if (n > 0)
        hipLaunchKernelGGL(sumsq_kernel, dim3(ed_grid_for(n, 256 * 8, 1024)), dim3(256), 0, s, g, n, sumsq_ws);
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, s, sumsq_ws, max_norm, pre_scale, coef,
                       norm_out);
    ED_CHECK_LAUNCH("grad_clip_coef");
    return ED_OK;
}

// ---------------------------------------------------------------------------------------------------
// Measurement aid (tools/rccl_footprint.py, tests/test_dp_gpu.py): a stand-in for what a collective library's
// resident kernels take from the chip while the launch-persistent recurrence kernels run - `workgroups` workgroups
// of 512 threads that hold >= 128 registers per lane and stream `bytes` of `buf` (read, add 0, write) `passes`
// times.  No 8-GPU node is available to the builder; this answers whether such a kernel on the auxiliary stream
// starves a recurrence launch into its bounded-spin give-up.
namespace {
__global__ __launch_bounds__(512) void footprint_kernel(float4* __restrict__ buf, long long n16, int passes) {
    float4 keep[32];                                   // 128 live registers per lane, as a ring kernel's staging
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 32; ++k) keep[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = 0; p < passes; ++p) {
        for (long long i = i0; i < n16; i += stride * 32) {
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const long long j = i + (long long)k * stride;
                if (j < n16) keep[k] = buf[j];
            }
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const long long j = i + (long long)k * stride;
                if (j < n16) {
                    float4 v = keep[k];
                    v.x += 0.f; v.y += 0.f; v.z += 0.f; v.w += 0.f;
                    buf[j] = v;
                }
            }
        }
    }
}
}  // namespace

extern "C" int edgedict_debug_footprint(float* buf, long long n_floats, int workgroups, int passes, void* stream) {
    ED_CHECK_ARG(buf && n_floats >= 4 && workgroups > 0 && passes > 0, "debug_footprint: bad arguments");
    hipLaunchKernelGGL(footprint_kernel, dim3(workgroups), dim3(512), 0, (hipStream_t)stream,
                       reinterpret_cast<float4*>(buf), n_floats / 4, passes);
    ED_CHECK_LAUNCH("footprint_kernel");
    return ED_OK;
}

