// RNN-T forward-backward loss for gfx950.
//
// Replaces the third-party warprnnt_pytorch.RNNTLoss operator used at
// rnnt/models.py:221,238 (reference).  Arithmetic = Graves 2012:
//   lp = log_softmax(z);  alpha/beta lattice recursions in log space;  cost = -alpha(T-1,U)-lp_blank
//   d cost / d z(t,u,v) = softmax(v)*exp(a+b-ll) - [v=blank]*exp(a+lp+beta(t+1,u)-ll)
//                                              - [v=y_{u+1}]*exp(a+lp+beta(t,u+1)-ll)
//
// Three HBM-/latency-shaped kernels, none of them GEMM-shaped (no MFMA here on purpose):
//   1. rnnt_lse_gather : one wave64 per lattice cell row (V logits), 16-byte coalesced loads,
//                        online max/sum, writes denominator + blank/label log-probs.   HBM-bound.
//   2. rnnt_alpha_beta : one WAVE per (utterance, direction); a lane owns ceil(U1/64) label columns and
//                        runs one row behind its left neighbour, whose hand-over arrives by a DPP
//                        shuffle (no LDS, no barrier), next row's log-probs requested a step ahead.
//                                                                              latency-bound.
//   3. rnnt_grad       : one wave64 per row again; reads logits once, writes grads once. HBM-bound.
#include "common.hpp"

namespace {

struct WsLayout {
    size_t cells;  // B*T*U1
    size_t off_denom, off_lpb, off_lpl, off_alpha, off_beta, off_ll, total;
};

inline WsLayout ws_layout(int B, int T, int U1) {
    WsLayout w;
    w.cells = (size_t)B * T * U1;
    const size_t cell_bytes = ((w.cells * sizeof(float) + 255) / 256) * 256;
    // alpha / beta / log-likelihoods are float64: the lattice values reach |ll| ~ 1e3 and
    // the gradient needs exp(alpha+beta-ll), i.e. the difference of three such numbers
    w.off_denom = 0;
    w.off_lpb = cell_bytes;
    w.off_lpl = 2 * cell_bytes;
    w.off_alpha = 3 * cell_bytes;
    w.off_beta = 5 * cell_bytes;
    w.off_ll = 7 * cell_bytes;
    w.total = 7 * cell_bytes + (((size_t)2 * B * sizeof(double) + 255) / 256) * 256;
    return w;
}

// ------------------------------------------------------------------ kernel 1
// grid-stride over rows; 4 waves per block, one row per wave per iteration.
template <typename T>
__global__ __launch_bounds__(256) void rnnt_lse_gather(
    const T* __restrict__ acts, const int32_t* __restrict__ labels,
    const int32_t* __restrict__ act_lens, const int32_t* __restrict__ label_lens, int B, int Tm,
    int U1, int V, int blank, float* __restrict__ denom, float* __restrict__ lpb,
    float* __restrict__ lpl, int vec_ok, const long long* __restrict__ pk_off) {
    constexpr int VEC = ElemIO<T>::VEC;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // grid (x, B): the workgroups of column b walk the VALID cells r = t (U_b + 1) + u of utterance b
    // (cells outside the box are never read) with 32-bit index arithmetic - three 64-bit divisions
    // per row were as much VALU work as the row itself
    const int b = blockIdx.y;
    const int Tb = min(act_lens[b], Tm), Ub = min(label_lens[b], U1 - 1);
    const int Wb = Ub + 1, nvalid = Tb * Wb;
    for (int r = blockIdx.x * 4 + wave; r < nvalid; r += gridDim.x * 4) {
        const int t = r / Wb, u = r - t * Wb;
        const long long row = ((long long)b * Tm + t) * U1 + u;
        // packed lattice: only the valid cells exist, utterance b starts at row pk_off[b] and its
        // rows are (U_b + 1) apart in t
        const long long arow = pk_off ? pk_off[b] + r : row;
        const T* z = acts + arow * (long long)V;
        float m = -INFINITY, s = 0.f;
        if (vec_ok) {
            for (int v = lane * VEC; v < V; v += 64 * VEC) {
                float x[VEC];
                ElemIO<T>::load_vec(z + v, x);
                float mx = x[0];
#pragma unroll
                for (int i = 1; i < VEC; ++i) mx = fmaxf(mx, x[i]);
                const float mn = fmaxf(m, mx);
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc += __expf(x[i] - mn);
                s = s * __expf(m - mn) + acc;
                m = mn;
            }
        } else {
            for (int v = lane; v < V; v += 64) {
                const float x = ElemIO<T>::load(z + v);
                const float mn = fmaxf(m, x);
                s = s * __expf(m - mn) + __expf(x - mn);
                m = mn;
            }
        }
        // combine the 64 (m, s) pairs
        const float M = wave_max(m);
        const float part = (m == -INFINITY) ? 0.f : s * __expf(m - M);
        const float S = wave_sum(part);
        const float lse = M + logf(S);
        if (lane == 0) {
            denom[row] = lse;
            lpb[row] = ElemIO<T>::load(z + blank) - lse;
            float l = 0.f;
            if (u < Ub) {
                const int y = labels[(long long)b * (U1 - 1) + u];
                l = ElemIO<T>::load(z + y) - lse;
            }
            lpl[row] = l;
        }
    }
}

// kernel 1 with the row maxima / exp-sums already reduced per 64-column slot by the logits product's
// epilogue (gemm_nt256.hip): half a wave per lattice row combines the `slots` pairs, one lane picks the
// blank / label logits out of the stored row.  Reads 8 * slots + 4 bytes per row instead of 2 V.
__global__ __launch_bounds__(256) void rnnt_lse_from_parts(
    const bf16_t* __restrict__ acts, const float2* __restrict__ parts, int slots,
    const int32_t* __restrict__ labels, const int32_t* __restrict__ act_lens,
    const int32_t* __restrict__ label_lens, int Tm, int U1, int V, int blank,
    float* __restrict__ denom, float* __restrict__ lpb, float* __restrict__ lpl,
    const long long* __restrict__ pk_off) {
    // ONE LANE per lattice row (no cross-lane reduction, the three outputs of consecutive rows leave as coalesced
    // stores), but the row's `slots` pairs (8 * slots contiguous bytes) reach the lane through LDS: a wave copies the
    // pairs of its next 64 (32) rows - one contiguous block of the packed lattice - with coalesced 16-byte loads and each
    // lane then reads its own row.  Read straight from global memory, lane by lane, the 64 lanes of a load touch 64
    // different lines of which 16 bytes are used; the line is gone from the CU's cache before the loop comes back for
    // the next 16: the PMC counters showed 1.08 GB fetched for 139 MB of pairs (0.20 ms at the HBM roof).
    // The arithmetic per row - pairs merged two at a time in slot order - is unchanged.
    constexpr int WAVE_F4 = 64 * 17;                       // float4 per wave: 64 rows x (16 + 1) or 32 rows x (33 + 1)
    __shared__ float4 stage[4][WAVE_F4];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Tb = max(0, min(act_lens[b], Tm)), Ub = max(0, min(label_lens[b], U1 - 1));   // as rnnt_alpha_beta
    const int Wb = Ub + 1, nvalid = Tb * Wb;
    const int q4 = slots >> 1;                             // 16-byte pieces per row (slots even on this path)
    const int RP = (slots & 1) ? 0 : (q4 <= 16 ? 64 : (q4 <= 33 ? 32 : 0));   // rows per wave and pass (0: direct loads)
    const int stride = q4 + 1;
    float4* sp = stage[wave];
    const int step = RP ? RP : 64;
    for (int r0 = (blockIdx.x * 4 + wave) * step; r0 < nvalid; r0 += gridDim.x * 4 * step) {
        const int nrows = min(step, nvalid - r0);
        const long long arow0 = pk_off[b] + r0;
        if (RP) {
            const float4* p4 = reinterpret_cast<const float4*>(parts + arow0 * slots);       // slots even: 16-byte aligned
            __builtin_amdgcn_wave_barrier();               // (the previous pass's reads of the stage are done)
            for (int i = lane; i < nrows * q4; i += 64) {
                const int rr = i / q4;
                sp[rr * stride + (i - rr * q4)] = p4[i];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (lane >= nrows) continue;
        const int r = r0 + lane;
        const int t = r / Wb, u = r - t * Wb;
        const long long row = ((long long)b * Tm + t) * U1 + u;
        const long long arow = arow0 + lane;
        float m = -INFINITY, sm = 0.f;
        int k = 0;
        if (RP) {
            for (; k + 1 < slots; k += 2) {
                const float4 p = sp[lane * stride + (k >> 1)];      // (max, sum) of slots k and k + 1
                const float nm = fmaxf(m, fmaxf(p.x, p.z));
                if (nm != -INFINITY) sm = sm * __expf(m - nm) + p.y * __expf(p.x - nm) + p.w * __expf(p.z - nm);
                m = nm;
            }
        } else {
            const float4* p4 = reinterpret_cast<const float4*>(parts + arow * slots);
            for (; (slots & 1) == 0 && k + 1 < slots; k += 2) {
                const float4 p = p4[k >> 1];
                const float nm = fmaxf(m, fmaxf(p.x, p.z));
                if (nm != -INFINITY) sm = sm * __expf(m - nm) + p.y * __expf(p.x - nm) + p.w * __expf(p.z - nm);
                m = nm;
            }
            for (; k < slots; ++k) {                   // odd slot counts: 8-byte loads
                const float2 p = parts[arow * slots + k];
                const float nm = fmaxf(m, p.x);
                if (nm != -INFINITY) sm = sm * __expf(m - nm) + p.y * __expf(p.x - nm);
                m = nm;
            }
        }
        const float lse = m + logf(sm);
        const bf16_t* z = acts + arow * (long long)V;
        denom[row] = lse;
        lpb[row] = bf16_to_f32(z[blank]) - lse;
        float l = 0.f;
        if (u < Ub) l = bf16_to_f32(z[labels[(long long)b * (U1 - 1) + u]]) - lse;
        lpl[row] = l;
    }
}

// log(exp(a)+exp(b)) with float64 carry: only the add/sub/max are fp64, the correction term
// log(1 + exp(-|a-b|)) in [0, ln 2] is evaluated in fp32 on the hardware transcendental units (v_exp_f32 /
// v_log_f32, ~1 ulp each: abs error ~1e-7, the same order as libm's log1pf(expf(.)) - whose ~100 dependent
// instructions were 90 % of a lattice step: the recurrence is a chain of T + U of these)
__device__ __forceinline__ double log_add64(double a, double b) {
    const double m = fmax(a, b);
    if (m == -(double)INFINITY) return m;
    const float d = (float)(fmin(a, b) - m);  // <= 0, may be -inf
    // 1 + x rounds to 1 below x ~ 6e-8 and carries 6e-8 of absolute rounding error wherever it is formed: a relative
    // error of 6e-4 in the correction term at x = 1e-4, 6e-6 at 1e-2.  Below 1e-2 the series x - x^2 / 2 + x^3 / 3
    // (truncation < x^4 / 4 = 2.5e-9 at the switch, where 1.f + x is already off by 6e-8) keeps what
    // `__logf(1.f + x)` drops - over a chain of T + U log-adds the dropped terms were a one-sided bias (ADVICE r4 / r5;
    // bounded by tests/test_rnnt_loss_gpu.py on a 1000 x 201 lattice)
    const float x = __expf(d);
    const float c = x < 1e-2f ? x * (1.f - x * (0.5f - x * (1.f / 3.f))) : __logf(1.f + x);
    return m + (double)c;
}

// ------------------------------------------------------------------ kernel 2
// ONE WAVE per (utterance, direction): blockIdx.x = 2*b + dir (dir 0: alpha, dir 1: beta), 64 threads.  Lane l owns
// the C = ceil(U1 / 64) consecutive label columns [C l, C l + C) (beta: counted from the right end) and walks them
// down the frames one row per step, one step behind lane l - 1: at step s it is in row s - l.  What a cell needs from
// the column to its left (alpha: a(t,u-1) + lp_label(t,u-1); beta: b(t,u+1)) was produced by the SAME lane one
// statement earlier or by lane l - 1 in the step before - a register handed up one lane with a DPP shuffle; what it
// needs from the row above is the lane's own value of the step before.  No LDS, no barrier: T + ceil(U1/C) - 1 steps
// of C dependent log-adds each (E6D2: 233 steps of 2, against 265 diagonals with a workgroup barrier and an LDS round
// trip each: 0.17 -> 0.03 ms).  Per cell the arithmetic and its operand order are those of the barrier kernel it
// replaces (fp64 carry, fp32 correction term of log_add64 above): alphas, betas and the likelihoods are bit-identical
// to that kernel built with the same log_add64 (the correction term itself changed in rounds 4 and 5).
template <int C>
__global__ __launch_bounds__(64) void rnnt_alpha_beta(const float* __restrict__ lpb, const float* __restrict__ lpl,
                                                      const int32_t* __restrict__ act_lens,
                                                      const int32_t* __restrict__ label_lens, int Tm, int U1,
                                                      double* __restrict__ alphas, double* __restrict__ betas,
                                                      double* __restrict__ ll) {
    // log-probabilities are requested D steps ahead of their use (a step is ~0.3 us of dependent arithmetic, an L2
    // round trip ~1 us: one step ahead the recurrence waited for its loads on every row - 0.30 instead of 0.17 ms)
    constexpr int D = C <= 2 ? 8 : (C <= 4 ? 4 : (C <= 8 ? 2 : 1));
    const int b = blockIdx.x >> 1;
    const int dir = blockIdx.x & 1;
    const int lane = threadIdx.x;
    // clamped exactly as rnnt_lse_gather / rnnt_grad clamp them: malformed lengths (the Python shim
    // rejects them, a raw C-ABI caller may not) cannot index past the [Tm, U1] slab; an empty utterance
    // (Tb <= 0) has likelihood 0 => cost +inf, never garbage
    const int Tb = max(0, min(act_lens[b], Tm)), Ub = max(0, min(label_lens[b], U1 - 1));
    if (Tb == 0) {
        if (lane == 0) ll[2 * b + dir] = -(double)INFINITY;
        return;
    }
    const long long base = (long long)b * Tm * U1;
    const int nlanes = (Ub + C) / C;                 // lanes that own a column <= Ub
    const int nsteps = Tb + nlanes - 1;
    const double NEG = -(double)INFINITY;
    double keep[C];                                  // alpha: a(t-1,u) + lp_blank(t-1,u);  beta: b(t+1,u)
#pragma unroll
    for (int c = 0; c < C; ++c) keep[c] = NEG;
    double out_last = NEG;                           // what the lane to the right needs from this lane's last step
    // column of slot c: alpha u = C lane + c; beta u = Ub - (C lane + c); row number r of this lane -> frame
    auto col = [&](int c) { return dir == 0 ? C * lane + c : Ub - (C * lane + c); };
    auto frame = [&](int r) { return dir == 0 ? r : Tb - 1 - r; };
    float qb[D][C], ql[D][C];                        // ring by STEP: slot s % D holds the row this lane is in at step s
    auto request = [&](int j, int r) {               // (static j)
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int u = col(c);
            float vb = 0.f, vl = 0.f;
            if (lane < nlanes && r >= 0 && r < Tb && u >= 0 && u <= Ub) {
                const long long idx = base + (long long)frame(r) * U1 + u;
                vb = lpb[idx];
                vl = lpl[idx];
            }
            qb[j][c] = vb;
            ql[j][c] = vl;
        }
    };
#pragma unroll
    for (int j = 0; j < D; ++j) request(j, j - lane);
    for (int s0 = 0; s0 < nsteps; s0 += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int s = s0 + j;
            if (s >= nsteps) break;
            const int r = s - lane;                  // rows done before this one
            const bool live = lane < nlanes && r >= 0 && r < Tb;
            // lane l - 1's hand-over of the step before (same row): a whole-wave shift by one lane as two DPP moves
            // (wave_shr:1 - __shfl_up goes through the LDS crossbar and waits for it on every step)
            double side = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(out_last), 0x138, 0xf, 0xf, false),
                                           __builtin_amdgcn_update_dpp(0, __double2loint(out_last), 0x138, 0xf, 0xf, false));
            if (lane == 0) side = NEG;
            float cb[C], cl[C];
#pragma unroll
            for (int c = 0; c < C; ++c) { cb[c] = qb[j][c]; cl[c] = ql[j][c]; }
            request(j, r + D);                       // the row of step s + D
            if (live) {
                const int t = frame(r);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const int u = col(c);
                    if (u < 0 || u > Ub) continue;
                    const long long idx = base + (long long)t * U1 + u;
                    if (dir == 0) {
                        // a(t,u) = lse(a(t-1,u) + lpb(t-1,u), a(t,u-1) + lpl(t,u-1))
                        const double a = (t == 0 && u == 0) ? 0.0 : log_add64(keep[c], side);
                        alphas[idx] = a;
                        keep[c] = a + cb[c];
                        side = (u < Ub) ? a + cl[c] : NEG;
                        if (t == Tb - 1 && u == Ub) ll[2 * b] = a + cb[c];
                    } else {
                        // b(t,u) = lse(b(t+1,u) + lpb(t,u), b(t,u+1) + lpl(t,u))
                        double bv;
                        if (t == Tb - 1 && u == Ub) {
                            bv = cb[c];
                        } else {
                            const double via_label = (u < Ub) ? side + cl[c] : NEG;
                            const double via_blank = (t < Tb - 1) ? keep[c] + cb[c] : NEG;
                            bv = log_add64(via_blank, via_label);
                        }
                        betas[idx] = bv;
                        keep[c] = bv;
                        side = bv;
                        if (t == 0 && u == 0) ll[2 * b + 1] = bv;
                    }
                }
                out_last = side;
            } else {
                out_last = NEG;
            }
        }
    }
}

// single block: costs[b] = -ll_alpha[b]; optionally reduced[0] = reduce_scale * sum_b costs[b]
__global__ __launch_bounds__(256) void rnnt_costs(const double* __restrict__ ll,
                                                  float* __restrict__ costs, int B,
                                                  float* __restrict__ reduced, float reduce_scale) {
    __shared__ float part[4];
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float c = (float)(-ll[2 * b]);
        costs[b] = c;
        acc += c;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && reduced) reduced[0] = reduce_scale * (part[0] + part[1] + part[2] + part[3]);
}

// ------------------------------------------------------------------ kernel 3
#ifndef ED_GRAD_OCC
#define ED_GRAD_OCC 1
#endif
template <typename T>
__global__ __launch_bounds__(256, ED_GRAD_OCC) void rnnt_grad(
    const T* __restrict__ acts, T* __restrict__ grads, const int32_t* __restrict__ labels,
    const int32_t* __restrict__ act_lens, const int32_t* __restrict__ label_lens, int B, int Tm,
    int U1, int V, int blank, const float* __restrict__ denom, const double* __restrict__ alphas,
    const double* __restrict__ betas, const double* __restrict__ ll, float scale_host,
    const float* __restrict__ scale_dev, int scale_stride, int vec_ok,
    const long long* __restrict__ pk_off, int b0) {
    constexpr int VEC = ElemIO<T>::VEC;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // grid (x, B), 32-bit index arithmetic (see rnnt_lse_gather).  Dense layout: every cell of the
    // [Tm, U1] slab of utterance b is written (zeros outside its box); packed: only the box exists.
    const int b = b0 + blockIdx.y;      // utterances [b0, b0 + gridDim.y) of the batch
    const int Tb = min(act_lens[b], Tm), Ub = min(label_lens[b], U1 - 1);
    const float scale = scale_host * (scale_dev ? scale_dev[(long long)b * scale_stride] : 1.f);
    const int Wb = pk_off ? Ub + 1 : U1, ncells = pk_off ? Tb * Wb : Tm * U1;
    for (int r = blockIdx.x * 4 + wave; r < ncells; r += gridDim.x * 4) {
        const int t = r / Wb, u = r - t * Wb;
        const long long row = ((long long)b * Tm + t) * U1 + u;
        const bool inside = (t < Tb && u <= Ub);
        const long long arow = pk_off ? pk_off[b] + r : row;
        const T* z = acts + arow * (long long)V;
        T* g = grads + arow * (long long)V;
        // the first (for V <= 64 * VEC * 4: the only) batch of logits is requested BEFORE the cell's alpha / beta /
        // denominator are: the row does not depend on them, and behind them it would start a second round trip
        constexpr int NB = 4;
        uint4 raw[NB];
        if (vec_ok && inside) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int v = (lane + 64 * j) * VEC;
                if (v < V) raw[j] = *reinterpret_cast<const uint4*>(z + v);
            }
        }
        float c_all = 0.f, c_blank = -INFINITY, c_label = -INFINITY;
        int y = -1;
        if (inside) {
            const double a = alphas[row], bt_ = betas[row];
            const double L = ll[2 * b];
            const float lse = denom[row];
            c_all = (float)(a + bt_ - L) - lse;  // exp(z + c_all) = softmax * exp(a+b-L)
            if (t < Tb - 1)
                c_blank = (float)(a + betas[row + U1] - L) - lse;
            else if (u == Ub)
                c_blank = (float)(a - L) - lse;
            if (u < Ub) {
                y = labels[(long long)b * (U1 - 1) + u];
                c_label = (float)(a + betas[row + 1] - L) - lse;
            }
        }
        if (vec_ok) {
            for (int v = lane * VEC, j = 0; v < V; v += 64 * VEC, ++j) {
                float o[VEC];
                if (inside) {
                    float x[VEC];
                    if (j < NB) {
                        // (static indices only: a run-time raw[j] would put the array in scratch)
                        const uint4 rv = j == 0 ? raw[0] : (j == 1 ? raw[1] : (j == 2 ? raw[2] : raw[3]));
                        ElemIO<T>::cvt_vec(rv, x);
                    } else {
                        ElemIO<T>::load_vec(z + v, x);
                    }
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        float gv = __expf(x[i] + c_all);
                        if (v + i == blank) gv -= __expf(x[i] + c_blank);
                        if (v + i == y) gv -= __expf(x[i] + c_label);
                        o[i] = gv * scale;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) o[i] = 0.f;
                }
                ElemIO<T>::store_vec(g + v, o);
            }
        } else {
            for (int v = lane; v < V; v += 64) {
                float gv = 0.f;
                if (inside) {
                    const float x = ElemIO<T>::load(z + v);
                    gv = __expf(x + c_all);
                    if (v == blank) gv -= __expf(x + c_blank);
                    if (v == y) gv -= __expf(x + c_label);
                    gv *= scale;
                }
                ElemIO<T>::store(g + v, gv);
            }
        }
    }
}

// rnnt_grad for the PACKED lattice that also leaves the column sums of the gradient matrix (fp32 values in front of the
// store's rounding) as one partial row per workgroup in colsum_parts[blockIdx.y * gridDim.x + blockIdx.x][V]: the joint's
// output-bias gradient is the column sum of this matrix, and a separate pass over its 2.2 GB (E6D2 bench batch) on the
// auxiliary stream beside the encoder's BPTT cost the step 0.4 ms (profiles/r6_colsum.txt).
// Work split: the workgroup's rows are those of rnnt_grad (groups of 4: r = 4 (blockIdx.x + k gridDim.x) + q), but wave w
// owns COLUMN SLICE w (64 * VEC columns: one 16-byte vector per lane) of all four rows of a group instead of one whole
// row - VEC accumulators per lane with static indices, four loads in flight as before.  (Tried first: one row per wave
// with 4 x VEC register accumulators, 1.09 instead of 0.88 ms; LDS ds_add_f32 accumulators, 5.8 ms.)
// Needs 16-byte aligned rows and V <= 4 * 64 * VEC (2048 in bf16, 1024 in f32); arithmetic per cell as in rnnt_grad.
template <typename T>
__global__ __launch_bounds__(256, ED_GRAD_OCC) void rnnt_grad_cs(
    const T* __restrict__ acts, T* __restrict__ grads, const int32_t* __restrict__ labels,
    const int32_t* __restrict__ act_lens, const int32_t* __restrict__ label_lens, int B, int Tm,
    int U1, int V, int blank, const float* __restrict__ denom, const double* __restrict__ alphas,
    const double* __restrict__ betas, const double* __restrict__ ll, float scale_host,
    const float* __restrict__ scale_dev, int scale_stride, const long long* __restrict__ pk_off,
    float* __restrict__ colsum_parts) {
    constexpr int VEC = ElemIO<T>::VEC;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int Tb = min(act_lens[b], Tm), Ub = min(label_lens[b], U1 - 1);
    const float scale = scale_host * (scale_dev ? scale_dev[(long long)b * scale_stride] : 1.f);
    const int Wb = Ub + 1, ncells = Tb * Wb;
    const int v = (wave * 64 + lane) * VEC;          // this lane's columns v .. v + VEC - 1 of every row
    const bool col_live = v < V;
    const double L = ll[2 * b];
    float cs[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) cs[i] = 0.f;
    for (int r0 = blockIdx.x * 4; r0 < ncells; r0 += gridDim.x * 4) {
        uint4 raw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (col_live && r0 + q < ncells) raw[q] = *reinterpret_cast<const uint4*>(acts + (pk_off[b] + r0 + q) * (long long)V + v);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + q;
            if (r >= ncells) break;
            const int t = r / Wb, u = r - t * Wb;
            const long long row = ((long long)b * Tm + t) * U1 + u;
            const double a = alphas[row], bt_ = betas[row];
            const float lse = denom[row];
            const float c_all = (float)(a + bt_ - L) - lse;
            float c_blank = -INFINITY, c_label = -INFINITY;
            int y = -1;
            if (t < Tb - 1)
                c_blank = (float)(a + betas[row + U1] - L) - lse;
            else if (u == Ub)
                c_blank = (float)(a - L) - lse;
            if (u < Ub) {
                y = labels[(long long)b * (U1 - 1) + u];
                c_label = (float)(a + betas[row + 1] - L) - lse;
            }
            if (col_live) {
                float x[VEC], o[VEC];
                ElemIO<T>::cvt_vec(raw[q], x);
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    float gv = __expf(x[i] + c_all);
                    if (v + i == blank) gv -= __expf(x[i] + c_blank);
                    if (v + i == y) gv -= __expf(x[i] + c_label);
                    o[i] = gv * scale;
                    cs[i] += o[i];
                }
                ElemIO<T>::store_vec(grads + (pk_off[b] + r) * (long long)V + v, o);
            }
        }
    }
    if (col_live) {
        float* out = colsum_parts + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * V + v;
#pragma unroll
        for (int i = 0; i < VEC; ++i) out[i] = cs[i];
    }
}

inline int check_common(int B, int T, int U1, int V, int blank, int dtype) {
    ED_CHECK_ARG(B > 0 && T > 0 && U1 > 0 && V > 0, "rnnt_loss: B,T,U1,V must be positive (got %d,%d,%d,%d)", B, T, U1, V);
    ED_CHECK_ARG(U1 <= 1024, "rnnt_loss: U+1 = %d exceeds the supported maximum of 1024", U1);
    ED_CHECK_ARG(blank >= 0 && blank < V, "rnnt_loss: blank %d outside [0,%d)", blank, V);
    ED_CHECK_ARG(dtype == ED_F32 || dtype == ED_BF16, "rnnt_loss: unsupported dtype code %d", dtype);
    return ED_OK;
}

}  // namespace

extern "C" size_t edgedict_rnnt_workspace_bytes(int B, int T, int U1) {
    if (B <= 0 || T <= 0 || U1 <= 0) return 0;
    return ws_layout(B, T, U1).total;
}

extern "C" const void* edgedict_rnnt_workspace_view(const void* workspace, int B, int T, int U1,
                                                     int which) {
    const WsLayout w = ws_layout(B, T, U1);
    const char* p = (const char*)workspace;
    switch (which) {
        case 0: return p + w.off_denom;
        case 1: return p + w.off_alpha;
        case 2: return p + w.off_beta;
        case 3: return p + w.off_ll;
        case 4: return p + w.off_lpb;
        case 5: return p + w.off_lpl;
    }
    return nullptr;
}

static int loss_forward(const void* acts, int acts_dtype, const int32_t* labels,
                        const int32_t* act_lens, const int32_t* label_lens, int B, int T, int U1,
                        int V, int blank, float* costs, float* reduced, float reduce_scale,
                        void* workspace, const long long* pk_off, void* stream_,
                        const float* lse_parts = nullptr, int lse_slots = 0) {
    if (int rc = check_common(B, T, U1, V, blank, acts_dtype)) return rc;
    ED_CHECK_ARG(acts && (labels || U1 == 1) && act_lens && label_lens && costs && workspace,
                 "rnnt_loss_forward: null pointer argument");
    ED_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "rnnt_loss_forward: workspace must be 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    const WsLayout w = ws_layout(B, T, U1);
    char* p = (char*)workspace;
    float* denom = (float*)(p + w.off_denom);
    float* lpb = (float*)(p + w.off_lpb);
    float* lpl = (float*)(p + w.off_lpl);
    double* alphas = (double*)(p + w.off_alpha);
    double* betas = (double*)(p + w.off_beta);
    double* ll = (double*)(p + w.off_ll);

    const size_t esz = acts_dtype == ED_F32 ? 4 : 2;
    const int vec_ok = ((V * esz) % 16 == 0) && (((uintptr_t)acts & 15) == 0);
    const dim3 grid1(ed_grid_for((long long)T * U1, 4, max(1, 256 * 16 / B)), B);
    if (lse_parts) {
        ED_CHECK_ARG(((uintptr_t)lse_parts & 15) == 0, "rnnt_loss_forward: lse_parts must be 16-byte aligned");
        ED_CHECK_ARG(acts_dtype == ED_BF16 && pk_off && lse_slots > 0,
                     "rnnt_loss_forward: log-sum-exp partials need bf16 logits on the packed lattice");
        const dim3 gridp(ed_grid_for((long long)T * U1, 256, max(1, 256 * 16 / B)), B);
        hipLaunchKernelGGL(rnnt_lse_from_parts, gridp, dim3(256), 0, stream, (const bf16_t*)acts,
                           (const float2*)lse_parts, lse_slots, labels, act_lens, label_lens, T, U1, V,
                           blank, denom, lpb, lpl, pk_off);
    } else if (acts_dtype == ED_F32)
        hipLaunchKernelGGL(rnnt_lse_gather<float>, grid1, dim3(256), 0, stream,
                           (const float*)acts, labels, act_lens, label_lens, B, T, U1, V, blank,
                           denom, lpb, lpl, vec_ok, pk_off);
    else
        hipLaunchKernelGGL(rnnt_lse_gather<bf16_t>, grid1, dim3(256), 0, stream,
                           (const bf16_t*)acts, labels, act_lens, label_lens, B, T, U1, V, blank,
                           denom, lpb, lpl, vec_ok, pk_off);
    ED_CHECK_LAUNCH("rnnt_lse_gather");

    // one wave per (utterance, direction), C = ceil(U1 / 64) label columns per lane (rounded up to a power of two)
    ED_CHECK_ARG(U1 <= 64 * 32, "rnnt_loss_forward: more than 2047 labels per utterance (U1 = %d)", U1);
#define ED_AB_LAUNCH(CC) hipLaunchKernelGGL(rnnt_alpha_beta<CC>, dim3(2 * B), dim3(64), 0, stream, lpb, lpl, act_lens, \
                                            label_lens, T, U1, alphas, betas, ll)
    const int cols = (U1 + 63) / 64;
    if (cols <= 1) ED_AB_LAUNCH(1);
    else if (cols <= 2) ED_AB_LAUNCH(2);
    else if (cols <= 4) ED_AB_LAUNCH(4);
    else if (cols <= 8) ED_AB_LAUNCH(8);
    else if (cols <= 16) ED_AB_LAUNCH(16);
    else ED_AB_LAUNCH(32);
#undef ED_AB_LAUNCH
    ED_CHECK_LAUNCH("rnnt_alpha_beta");
    hipLaunchKernelGGL(rnnt_costs, dim3(1), dim3(256), 0, stream, ll, costs, B, reduced,
                       reduce_scale);
    ED_CHECK_LAUNCH("rnnt_costs");
    return ED_OK;
}

extern "C" int edgedict_rnnt_loss_forward(const void* acts, int acts_dtype, const int32_t* labels,
                                          const int32_t* act_lens, const int32_t* label_lens,
                                          int B, int T, int U1, int V, int blank, float* costs,
                                          float* reduced, float reduce_scale, void* workspace,
                                          void* stream_) {
    return loss_forward(acts, acts_dtype, labels, act_lens, label_lens, B, T, U1, V, blank, costs,
                        reduced, reduce_scale, workspace, nullptr, stream_);
}

extern "C" int edgedict_rnnt_loss_forward_packed(const void* acts, int acts_dtype,
                                                 const int32_t* labels, const int32_t* act_lens,
                                                 const int32_t* label_lens,
                                                 const long long* row_offsets, int B, int T, int U1,
                                                 int V, int blank, float* costs, float* reduced,
                                                 float reduce_scale, void* workspace, void* stream_) {
    ED_CHECK_ARG(row_offsets, "rnnt_loss_forward_packed: null row_offsets");
    return loss_forward(acts, acts_dtype, labels, act_lens, label_lens, B, T, U1, V, blank, costs,
                        reduced, reduce_scale, workspace, row_offsets, stream_);
}

extern "C" int edgedict_rnnt_loss_forward_packed_parts(const void* acts, const int32_t* labels,
                                                       const int32_t* act_lens,
                                                       const int32_t* label_lens,
                                                       const long long* row_offsets, int B, int T,
                                                       int U1, int V, int blank, float* costs,
                                                       float* reduced, float reduce_scale,
                                                       void* workspace, const float* lse_parts,
                                                       int lse_slots, void* stream_) {
    ED_CHECK_ARG(row_offsets && lse_parts, "rnnt_loss_forward_packed_parts: null pointer");
    ED_CHECK_ARG(lse_slots == (V + 63) / 64, "rnnt_loss_forward_packed_parts: lse_slots must be ceil(V / 64)");
    return loss_forward(acts, ED_BF16, labels, act_lens, label_lens, B, T, U1, V, blank, costs,
                        reduced, reduce_scale, workspace, row_offsets, stream_, lse_parts, lse_slots);
}

// workgroups per utterance of rnnt_grad (grid.x; grid.y = utterances)
static int grad_grid_x(int B, int T, int U1) { return ed_grid_for((long long)T * U1, 4, max(1, 256 * 16 / B)); }

static int loss_backward(const void* acts, int acts_dtype, void* grads, const int32_t* labels,
                         const int32_t* act_lens, const int32_t* label_lens, int B, int T, int U1,
                         int V, int blank, const void* workspace, float grad_scale_host,
                         const float* grad_scale_dev, int grad_scale_stride,
                         const long long* pk_off, void* stream_, int b0 = 0, int nb = -1,
                         float* colsum_parts = nullptr) {
    if (int rc = check_common(B, T, U1, V, blank, acts_dtype)) return rc;
    if (nb < 0) nb = B - b0;
    ED_CHECK_ARG(b0 >= 0 && nb >= 0 && b0 + nb <= B, "rnnt_loss_backward: utterance range [%d, %d) outside the batch of %d", b0, b0 + nb, B);
    if (nb == 0) return ED_OK;
    ED_CHECK_ARG(acts && grads && (labels || U1 == 1) && act_lens && label_lens && workspace,
                 "rnnt_loss_backward: null pointer argument");
    hipStream_t stream = (hipStream_t)stream_;
    const WsLayout w = ws_layout(B, T, U1);
    const char* p = (const char*)workspace;
    const float* denom = (const float*)(p + w.off_denom);
    const double* alphas = (const double*)(p + w.off_alpha);
    const double* betas = (const double*)(p + w.off_beta);
    const double* ll = (const double*)(p + w.off_ll);
    const size_t esz = acts_dtype == ED_F32 ? 4 : 2;
    const int vec_ok = ((V * esz) % 16 == 0) && (((uintptr_t)acts & 15) == 0) &&
                       (((uintptr_t)grads & 15) == 0);
    const dim3 grid(grad_grid_x(B, T, U1), nb);
    if (colsum_parts) {
        const int cs_width = 4 * 64 * (acts_dtype == ED_F32 ? 4 : 8);      // four column slices of 64 lanes x 16 bytes
        ED_CHECK_ARG(vec_ok && V <= cs_width && pk_off && b0 == 0 && nb == B,
                     "rnnt_loss_backward: fused column sums need the packed lattice, 16-byte aligned rows and V <= %d (got V = %d)",
                     cs_width, V);
        if (acts_dtype == ED_F32)
            hipLaunchKernelGGL(rnnt_grad_cs<float>, grid, dim3(256), 0, stream, (const float*)acts, (float*)grads, labels,
                               act_lens, label_lens, B, T, U1, V, blank, denom, alphas, betas, ll, grad_scale_host,
                               grad_scale_dev, grad_scale_stride, pk_off, colsum_parts);
        else
            hipLaunchKernelGGL(rnnt_grad_cs<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)acts, (bf16_t*)grads,
                               labels, act_lens, label_lens, B, T, U1, V, blank, denom, alphas, betas, ll,
                               grad_scale_host, grad_scale_dev, grad_scale_stride, pk_off, colsum_parts);
        ED_CHECK_LAUNCH("rnnt_grad_cs");
        return ED_OK;
    }
    if (acts_dtype == ED_F32)
        hipLaunchKernelGGL(rnnt_grad<float>, grid, dim3(256), 0, stream, (const float*)acts,
                           (float*)grads, labels, act_lens, label_lens, B, T, U1, V, blank, denom,
                           alphas, betas, ll, grad_scale_host, grad_scale_dev, grad_scale_stride,
                           vec_ok, pk_off, b0);
    else
        hipLaunchKernelGGL(rnnt_grad<bf16_t>, grid, dim3(256), 0, stream,
                           (const bf16_t*)acts, (bf16_t*)grads, labels, act_lens, label_lens, B, T,
                           U1, V, blank, denom, alphas, betas, ll, grad_scale_host, grad_scale_dev,
                           grad_scale_stride, vec_ok, pk_off, b0);
    ED_CHECK_LAUNCH("rnnt_grad");
    return ED_OK;
}

extern "C" int edgedict_rnnt_loss_backward(const void* acts, int acts_dtype, void* grads,
                                           const int32_t* labels, const int32_t* act_lens,
                                           const int32_t* label_lens, int B, int T, int U1, int V,
                                           int blank, const void* workspace, float grad_scale_host,
                                           const float* grad_scale_dev, int grad_scale_stride,
                                           void* stream_) {
    return loss_backward(acts, acts_dtype, grads, labels, act_lens, label_lens, B, T, U1, V, blank,
                         workspace, grad_scale_host, grad_scale_dev, grad_scale_stride, nullptr, stream_);
}

extern "C" int edgedict_rnnt_loss_backward_packed(const void* acts, int acts_dtype, void* grads,
                                                  const int32_t* labels, const int32_t* act_lens,
                                                  const int32_t* label_lens,
                                                  const long long* row_offsets, int B, int T, int U1,
                                                  int V, int blank, const void* workspace,
                                                  float grad_scale_host, const float* grad_scale_dev,
                                                  int grad_scale_stride, void* stream_) {
    ED_CHECK_ARG(row_offsets, "rnnt_loss_backward_packed: null row_offsets");
    return loss_backward(acts, acts_dtype, grads, labels, act_lens, label_lens, B, T, U1, V, blank,
                         workspace, grad_scale_host, grad_scale_dev, grad_scale_stride, row_offsets,
                         stream_);
}

extern "C" int edgedict_rnnt_grad_colsum_rows(int acts_dtype, int B, int T, int U1, int V) {
    if (B <= 0 || T <= 0 || U1 <= 0 || V <= 0 || (acts_dtype != ED_F32 && acts_dtype != ED_BF16)) return 0;
    const int esz = acts_dtype == ED_F32 ? 4 : 2;
    if ((V * esz) % 16 != 0 || V > 4 * 64 * (16 / esz)) return 0;
    return grad_grid_x(B, T, U1) * B;
}

extern "C" int edgedict_rnnt_loss_backward_packed_colsum(const void* acts, int acts_dtype, void* grads,
                                                         const int32_t* labels, const int32_t* act_lens,
                                                         const int32_t* label_lens,
                                                         const long long* row_offsets, int B, int T, int U1,
                                                         int V, int blank, const void* workspace,
                                                         float grad_scale_host, const float* grad_scale_dev,
                                                         int grad_scale_stride, float* colsum_parts, void* stream_) {
    ED_CHECK_ARG(row_offsets && colsum_parts, "rnnt_loss_backward_packed_colsum: null pointer");
    return loss_backward(acts, acts_dtype, grads, labels, act_lens, label_lens, B, T, U1, V, blank,
                         workspace, grad_scale_host, grad_scale_dev, grad_scale_stride, row_offsets,
                         stream_, 0, -1, colsum_parts);
}

extern "C" int edgedict_rnnt_loss_backward_packed_range(const void* acts, int acts_dtype, void* grads,
                                                        const int32_t* labels, const int32_t* act_lens,
                                                        const int32_t* label_lens,
                                                        const long long* row_offsets, int B, int T, int U1,
                                                        int V, int blank, const void* workspace,
                                                        float grad_scale_host, const float* grad_scale_dev,
                                                        int grad_scale_stride, int b0, int nb, void* stream_) {
    ED_CHECK_ARG(row_offsets, "rnnt_loss_backward_packed_range: null row_offsets");
    return loss_backward(acts, acts_dtype, grads, labels, act_lens, label_lens, B, T, U1, V, blank,
                         workspace, grad_scale_host, grad_scale_dev, grad_scale_stride, row_offsets,
                         stream_, b0, nb);
}
