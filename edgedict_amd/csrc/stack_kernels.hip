// Kernels of the layer-pipelined encoder stack (bf16 throughput mode).
//
// Reference arithmetic: ResLayerNormLSTM.forward rnnt/models.py:55-75 (nn.LSTM per layer, residual
// add for layers > 0, LayerNorm, TimeReduction rnnt/models.py:21-29) and its autograd.  PyTorch
// gate order i,f,g,o; c_t = f*c_{t-1} + i*g; h_t = o*tanh(c_t).
//
// Why these kernels exist next to lstm_fast.hip: a dependent kernel boundary costs ~1.7 us on
// MI355X and one LSTM step of ONE layer cannot use more than ~64 CUs' worth of L2 bandwidth, so a
// kernel-per-step recurrence leaves the chip idle (7 us/step, 1606 steps).  Here ONE launch carries
// one time step of EVERY layer that is currently runnable (the layers run as a skewed wavefront,
// see encoder_stack.hip) plus the LayerNorms of the frames finished by the previous launch:
//
//   forward step tile : 64 batch rows x 16 hidden units x 4 gates, K = H split over the 4 waves
//                       -> 64 workgroups per layer-step, 256 KB of L2 reads each (W_hh slice 128 KB
//                       + all of h_{t-1} 128 KB), operands in MFMA fragment order (lstm_fast.hip)
//   backward step tile: 32 rows x 32 units, K = 4H split over the waves -> 64 workgroups, 512 KB
//   norm              : one wave per output row (pair mean of two frames under time reduction)
//
// All activations are TIME-MAJOR ([T, B, *]) so one step touches contiguous rows and a chunk of
// frames is a contiguous GEMM operand; gate columns are interleaved (ed_gate_col) so the 4 gates
// of 16 units are one 128-byte line.  Partial sums of the K split are handed to the wave that
// owns the tile through LDS lane-wise (conflict-free 16-byte accesses); every global access of
// the epilogue is a 16-byte vector staged through LDS.
#include "stack_kernels.hpp"

#ifndef ED_STEP_PRIO
#define ED_STEP_PRIO 3   // s_setprio of the recurrence kernels' waves (0..3): they are latency-critical, the products
#endif                   // that share their CUs are not (measured: backward pass 11.9 -> 11.5 ms, forward unchanged)
#ifndef ED_STACK_DBG
#define ED_STACK_DBG 0   // tools/stack_probe.hip builds ablation variants with this mask:
#endif                   // 1 no operand loads/MFMA, 2 no stores, 4 cheap activations, 8 no staging loads

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ bf16x8_t ldfrag(const bf16_t* p) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    return *reinterpret_cast<bf16x8_t*>(&v);
}
__device__ __forceinline__ bf16x8_t zfrag() {
    uint4 v = make_uint4(0, 0, 0, 0);
    return *reinterpret_cast<bf16x8_t*>(&v);
}
// v_exp/v_rcp based activations (abs. error ~1e-7; the results are rounded to bf16 anyway)
__device__ __forceinline__ float fsigmoid(float x) { return __fdividef(1.f, 1.f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) {
    const float xc = fminf(fmaxf(x, -15.f), 15.f);
    return 1.f - __fdividef(2.f, 1.f + __expf(2.f * xc));
}

// =====================================================================================
// forward
// =====================================================================================
#ifndef ED_FCH
#define ED_FCH 2
#endif
#ifndef ED_FWD_OCC
#define ED_FWD_OCC 2
#endif
constexpr int FCH = ED_FCH;   // k-steps per register buffer: 2 x (4 A + 4 W) x 16 B = 256 B / lane, two buffers

struct __attribute__((aligned(16))) FwdShared {
    float4 hand[4][3][4][64];   // [source wave][destination slot][gate][lane]   48 KB
    bf16_t g[64][72];           // pre-activations in, gates out (64 cols + 8 pad)  9 KB
    float c[64][20];            // c_{t-1} in, c_t out (16 + 4 pad)                 5 KB
    bf16_t h[64][24];           // h_t (16 + 8 pad)                                 3 KB
};

__device__ __forceinline__ void fwd_step_role(const EdFwdStep& p, int ub, int rg, int B, int H,
                                              FwdShared& sh) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int KS = H >> 5;
    const int per = (KS + 3) >> 2;
    const int ks_beg = wave * per, ks_end = min(KS, ks_beg + per);
    const int MT = (B + 15) >> 4, mt0 = rg * 4, row0 = rg * 64;
    const long long H4 = 4ll * H;

    // ---- epilogue operands: requested first, parked in registers until the MFMAs are done
    uint4 gin[2];
    float4 cin;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + 256 * i, r = id >> 3, part = id & 7, b = row0 + r;
        gin[i] = make_uint4(0, 0, 0, 0);
        if (b < B && !(ED_STACK_DBG & 8)) gin[i] = *reinterpret_cast<const uint4*>(p.G_t + b * H4 + ub * 64 + part * 8);
    }
    {
        const int r = tid >> 2, part = tid & 3, b = row0 + r;
        cin = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < B && !(ED_STACK_DBG & 8)) cin = *reinterpret_cast<const float4*>(p.C_prev + (long long)b * H + ub * 16 + part * 4);
    }

    // ---- W_hh h_{t-1}: this wave's K quarter, all 4 row tiles x 4 gates
    f32x4_t acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[m][g] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bf16_t* abase = p.hfrag_in + lane * 8;
    const bf16_t* wbase = p.Wfrag + ((long long)ub * 4 * KS * 64 + lane) * 8;
    // two register buffers of FCH k-steps: the loads of one are in flight under the MFMAs of the other
    bf16x8_t a0[FCH][4], w0[FCH][4], a1[FCH][4], w1[FCH][4];
    auto load = [&](bf16x8_t (&a)[FCH][4], bf16x8_t (&w)[FCH][4], int ks0) {
#pragma unroll
        for (int i = 0; i < FCH; ++i) {
            const int ks = min(ks0 + i, ks_end - 1);   // clamp: duplicates are masked in mma()
#pragma unroll
            for (int m = 0; m < 4; ++m)
                // (row tiles past the batch: a duplicate of the last tile instead of a branch - their rows are never
                // stored, and straight-line loads let the compiler count its waits, see bwd_step_role)
                a[i][m] = ldfrag(abase + ((long long)ks * MT + min(mt0 + m, MT - 1)) * 512);
#pragma unroll
            for (int g = 0; g < 4; ++g) w[i][g] = ldfrag(wbase + ((long long)ks * 4 + g) * 512);
        }
    };
    auto mma = [&](bf16x8_t (&a)[FCH][4], bf16x8_t (&w)[FCH][4], int ks0) {
#pragma unroll
        for (int i = 0; i < FCH; ++i) {
            if (ks0 + i < ks_end) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][m], w[i][g], acc[m][g], 0, 0, 0);
            }
        }
    };
    if (ks_beg < ks_end && !(ED_STACK_DBG & 1)) {
        // steady state WITHOUT conditional loads: behind an `if (...) load(...)` the compiler's wait counts are
        // those of the path that did not load (vmcnt(13) where 29 were in flight) - the second buffer then never
        // overlapped the first.  The conditions live in the loop bounds and in a peeled tail instead.
        load(a0, w0, ks_beg);
        int ks0 = ks_beg;
        for (; ks0 + 2 * FCH < ks_end; ks0 += 2 * FCH) {
            load(a1, w1, ks0 + FCH);
            __builtin_amdgcn_sched_barrier(0);      // the whole buffer is requested before the other one is consumed
            mma(a0, w0, ks0);
            __builtin_amdgcn_sched_barrier(0);
            load(a0, w0, ks0 + 2 * FCH);
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, w1, ks0 + FCH);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ks0 + FCH < ks_end) {
            load(a1, w1, ks0 + FCH);
            mma(a0, w0, ks0);
            mma(a1, w1, ks0 + FCH);
        } else {
            mma(a0, w0, ks0);
        }
    }

    // ---- hand the partial tiles to their owners (wave m owns row tile m), park the operands
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (m != wave) {
            const int slot = m < wave ? m : m - 1;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                sh.hand[wave][slot][g][lane] = make_float4(acc[m][g][0], acc[m][g][1], acc[m][g][2], acc[m][g][3]);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + 256 * i, r = id >> 3, part = id & 7;
        *reinterpret_cast<uint4*>(&sh.g[r][part * 8]) = gin[i];
    }
    *reinterpret_cast<float4*>(&sh.c[tid >> 2][(tid & 3) * 4]) = cin;
    __syncthreads();

    f32x4_t mine[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
        mine[g] = wave == 0 ? acc[0][g] : wave == 1 ? acc[1][g] : wave == 2 ? acc[2][g] : acc[3][g];
#pragma unroll
    for (int src = 0; src < 4; ++src) {
        if (src != wave) {
            const int slot = wave < src ? wave : wave - 1;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v = sh.hand[src][slot][g][lane];
                mine[g][0] += v.x; mine[g][1] += v.y; mine[g][2] += v.z; mine[g][3] += v.w;
            }
        }
    }

    // ---- cell update: lane owns unit u of rows wave*16 + (lane>>4)*4 + q
    const int u = lane & 15, rbase = wave * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rl = rbase + q;
        float ig, fg, gg, og, c, h;
        if (ED_STACK_DBG & 4) {
            ig = bf16_to_f32(sh.g[rl][u]) + mine[0][q]; fg = bf16_to_f32(sh.g[rl][16 + u]) + mine[1][q];
            gg = bf16_to_f32(sh.g[rl][32 + u]) + mine[2][q]; og = bf16_to_f32(sh.g[rl][48 + u]) + mine[3][q];
            c = fg * sh.c[rl][u] + ig * gg;
            h = og * c;
        } else {
            ig = fsigmoid(bf16_to_f32(sh.g[rl][u]) + mine[0][q]);
            fg = fsigmoid(bf16_to_f32(sh.g[rl][16 + u]) + mine[1][q]);
            gg = ftanh(bf16_to_f32(sh.g[rl][32 + u]) + mine[2][q]);
            og = fsigmoid(bf16_to_f32(sh.g[rl][48 + u]) + mine[3][q]);
            c = fg * sh.c[rl][u] + ig * gg;
            h = og * ftanh(c);
        }
        sh.g[rl][u] = f32_to_bf16(ig);
        sh.g[rl][16 + u] = f32_to_bf16(fg);
        sh.g[rl][32 + u] = f32_to_bf16(gg);
        sh.g[rl][48 + u] = f32_to_bf16(og);
        sh.c[rl][u] = c;
        sh.h[rl][u] = f32_to_bf16(h);
    }
    __syncthreads();

    // ---- 1024 16-byte stores: gates 512, Y 128, fragment image 128, c 256
#pragma unroll
    for (int i = 0; i < ((ED_STACK_DBG & 2) ? 0 : 4); ++i) {
        const int task = tid + 256 * i;
        if (task < 512) {
            const int r = task >> 3, part = task & 7, b = row0 + r;
            if (b < B)
                *reinterpret_cast<uint4*>(p.G_t + b * H4 + ub * 64 + part * 8) =
                    *reinterpret_cast<const uint4*>(&sh.g[r][part * 8]);
        } else if (task < 640) {
            const int id = task - 512, r = id >> 1, hf = id & 1, b = row0 + r;
            if (b < B)
                *reinterpret_cast<uint4*>(p.Y_t + (long long)b * H + ub * 16 + hf * 8) =
                    *reinterpret_cast<const uint4*>(&sh.h[r][hf * 8]);
        } else if (task < 768) {
            const int id = task - 640, m = id >> 5, kg2 = (id >> 4) & 1, r16 = id & 15;
            const int mt = mt0 + m;
            if (mt < MT && p.hfrag_out) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (mt * 16 + r16 < B) v = *reinterpret_cast<const uint4*>(&sh.h[m * 16 + r16][kg2 * 8]);
                const int ks = ub >> 1, kg = (ub & 1) * 2 + kg2;
                *reinterpret_cast<uint4*>(p.hfrag_out + (((long long)ks * MT + mt) * 64 + kg * 16 + r16) * 8) = v;
            }
        } else {
            const int id = task - 768, r = id >> 2, part = id & 3, b = row0 + r;
            if (b < B)
                *reinterpret_cast<float4*>(p.C_t + (long long)b * H + ub * 16 + part * 4) =
                    *reinterpret_cast<const float4*>(&sh.c[r][part * 4]);
        }
    }
}

// sum / centred sum of squares of one row of (y + r), 16-byte loads; rows are re-read from L1
__device__ __forceinline__ float nrow_sum(const bf16_t* y, const bf16_t* r, int H, int lane) {
    float s = 0.f;
    for (int c = lane * 8; c < H; c += 512) {
        float a[8];
        ElemIO<bf16_t>::load_vec(y + c, a);
        if (r) {
            float b[8];
            ElemIO<bf16_t>::load_vec(r + c, b);
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] += b[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += a[i];
    }
    return wave_sum(s);
}
__device__ __forceinline__ float nrow_sqdev(const bf16_t* y, const bf16_t* r, float mean, int H,
                                            int lane) {
    float s = 0.f;
    for (int c = lane * 8; c < H; c += 512) {
        float a[8];
        ElemIO<bf16_t>::load_vec(y + c, a);
        if (r) {
            float b[8];
            ElemIO<bf16_t>::load_vec(r + c, b);
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] += b[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float d = a[i] - mean;
            s += d * d;
        }
    }
    return wave_sum(s);
}

__device__ __forceinline__ void fwd_norm_role(const EdFwdNorm& p, int rb, int B, int H, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = rb * 4 + wave;
    if (b >= B) return;
    const bf16_t* y[2] = {p.y0 + (long long)b * H, p.y1 ? p.y1 + (long long)b * H : nullptr};
    const bf16_t* r[2] = {p.r0 ? p.r0 + (long long)b * H : nullptr,
                          p.r1 ? p.r1 + (long long)b * H : nullptr};
    float mean[2] = {0.f, 0.f}, rstd[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (!y[k]) continue;
        const float m = nrow_sum(y[k], r[k], H, lane) / (float)H;
        const float var = nrow_sqdev(y[k], r[k], m, H, lane) / (float)H;
        mean[k] = m;
        rstd[k] = rsqrtf(var + eps);
    }
    if (lane == 0) {
        p.mean0[b] = mean[0];
        p.rstd0[b] = rstd[0];
        if (y[1]) {
            p.mean1[b] = mean[1];
            p.rstd1[b] = rstd[1];
        }
    }
    bf16_t* out = p.out + (long long)b * p.out_stride;
    for (int c = lane * 8; c < H; c += 512) {
        float o[8], gm[8], bt[8];
        const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + c);
        const float4 g1 = *reinterpret_cast<const float4*>(p.gamma + c + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(p.beta + c);
        const float4 b1 = *reinterpret_cast<const float4*>(p.beta + c + 4);
        gm[0] = g0.x; gm[1] = g0.y; gm[2] = g0.z; gm[3] = g0.w;
        gm[4] = g1.x; gm[5] = g1.y; gm[6] = g1.z; gm[7] = g1.w;
        bt[0] = b0.x; bt[1] = b0.y; bt[2] = b0.z; bt[3] = b0.w;
        bt[4] = b1.x; bt[5] = b1.y; bt[6] = b1.z; bt[7] = b1.w;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (!y[k]) continue;
            float a[8];
            ElemIO<bf16_t>::load_vec(y[k] + c, a);
            if (r[k]) {
                float rr[8];
                ElemIO<bf16_t>::load_vec(r[k] + c, rr);
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] += rr[i];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] += (a[i] - mean[k]) * rstd[k] * gm[i] + bt[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] *= p.scale;
        if (p.drop_thresh) {
            // the reference's Dropout sees the LayerNorm's output as a tensor: round first, then mask and scale
            const long long i0 = ((long long)b * p.drop_T + p.tau) * H + c;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                o[i] = ed_drop_keep(p.drop_seed, i0 + i, p.drop_thresh) ? bf16_to_f32(f32_to_bf16(o[i])) * p.drop_scale : 0.f;
        }
        ElemIO<bf16_t>::store_vec(out + c, o);
    }
}

// ---- flag wait instead of a stream wait.  A step that opens a chunk needs the chunk's side-stream product.
// Ordering the recurrence stream behind it with hipStreamWaitEvent costs that stream 5-15 us per wait even
// when the event is long complete (~135 waits per pass); instead the side stream sets a word
// after the product (ed_stack_set_flag) and the workgroups of that one slot poll it - satisfied on the first
// poll in the normal case, because the scheduler opens a chunk only `margin` launches after its product
// was enqueued.  The producer's results were written back at ITS kernel end, before the flag kernel ran;
// this kernel has not touched those lines before the poll succeeds (caches are invalidated at kernel
// start), and the acquire fence covers the rest.  Bounded: a give-up code goes to the host-visible word.
__device__ __forceinline__ void soft_wait(const unsigned* flag, unsigned* err, unsigned code) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1u << 22)) {            // ~ seconds
                if (err) atomicCAS(err, 0u, code);
                break;
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// One lane waits until every counter[i] >= target[i] (the launch-persistent forward's arrival counters): placed
// on a side stream in front of the LayerNorm of the frames a recurrence launch is still working on, it orders
// that stream behind the steps WITHOUT an event record between the recurrence stream's launches (each record
// cost ~3.5 us of inter-launch gap there).  What the waited-for steps published (write-through, drained before
// the arrive) is in memory when the counter shows them; the kernels behind this one start with clean caches.
constexpr int WAIT_COUNTERS_MAX = ED_STACK_MAX_SLOTS;
struct WaitCountersArgs {
    const unsigned* counter[WAIT_COUNTERS_MAX];
    unsigned target[WAIT_COUNTERS_MAX];
    int n;
    unsigned* err;
};
__global__ void stack_wait_counters_kernel(WaitCountersArgs a) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < a.n; ++i) {
        unsigned spins = 0;
        while (__hip_atomic_load(a.counter[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.target[i]) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1u << 22)) {
                if (a.err) atomicCAS(a.err, 0u, 800u + i);
                return;
            }
        }
    }
}

__global__ void stack_set_flag_kernel(unsigned* flag) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256, ED_FWD_OCC) void stack_fwd_kernel(EdFwdLaunch L) {
    __shared__ FwdShared sh;
    // measurement mode (edgedict_stack_time_launches): first workgroup start / last workgroup end of this
    // launch on the constant 100 MHz clock - the kernel's own duration, as a profiler's begin/end sees it
    if (L.stamp && threadIdx.x == 0) atomicMin(&L.stamp[0], wall_clock64());
    const int UB = L.H >> 4, RG = (L.B + 63) >> 6;
    const int nsb = L.nstep * UB * RG;
    int bid = blockIdx.x;
    if (bid < nsb) {
        const int slot = bid / (UB * RG), rem = bid - slot * UB * RG;
        if (L.step[slot].wait_flag) soft_wait(L.step[slot].wait_flag, L.err, 500u + slot);
        fwd_step_role(L.step[slot], rem % UB, rem / UB, L.B, L.H, sh);
    } else {
        bid -= nsb;
        const int RB = (L.B + 3) >> 2;
        fwd_norm_role(L.norm[bid / RB], bid % RB, L.B, L.H, L.eps);
    }
    if (L.stamp) {
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(&L.stamp[1], wall_clock64());
    }
}

// =====================================================================================
// forward, launch-persistent (EdLpwLaunch): `nsteps` consecutive time steps per launch.
//
// Why: with one launch per time step a step costs a dependent kernel boundary (1.7-3.6 us) plus the
// re-fetch of the workgroup's W_hh slice (128 KB) on top of the h image (128 KB): 11-12 us of kernel,
// 14.6 us of period, of which the recurrence itself needs the h exchange, 128 MFMA per wave and the cell
// update.  Here a workgroup owns the same (layer, 16 hidden units, 64 rows) for `nsteps` steps:
//   * its W_hh slice is loaded ONCE per launch into 128 registers per lane (wave w: k-quarter w, all 4 gates),
//   * its cell state c stays in LDS between steps,
//   * the only per-step global traffic on the dependency chain is the h fragment image: written with
//     write-through (sc1) stores, every writing wave drains, ONE lane bumps the layer's arrival counter;
//     readers poll that counter (one lane, relaxed) and read the image with L2-served (sc1) loads - the
//     recipe of cdna_hip_programming.md G16,
//   * everything a LATER kernel reads (gates for the backward pass, h rows for the LayerNorm, c rows) leaves
//     after the publish, off the chain, with plain stores.
// The arithmetic - MFMA order per wave, order of the cross-wave sum, cell math - is that of fwd_step_role,
// so the results are bit-identical to the launch-per-step kernel (tests/test_lpw_gpu.py).
// Co-residency: the grid is at most one workgroup per CU (the scheduler limits the slots); a workgroup that
// is not resident yet just delays its layer's counter; every spin is bounded (give-up code 700 + slot).
// =====================================================================================
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
constexpr int LPW_PER = 8;     // k-steps of 32 per wave held in registers: H <= 1024

__device__ __forceinline__ __amdgpu_buffer_rsrc_t lpw_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (int)bytes, 0x00020000);
}

// (Round 4, measured and removed: the workgroup's four row tiles as 2 / 4 sub-batches with their own arrival counters,
// advanced alternately with deferred arrivals - bit-identical, but 5.1 -> 5.9 / 8.2 ms per forward pass: a gather of half
// the rows takes 2.3 of the 3.2 us of the full one, and an arrival deferred to the next sub-iteration comes too late.)
//
// Data polling (DP, the default): the hand-off through a counter is three latencies in a row - the publisher waits
// for its write-through stores to be acknowledged (1.3 us), bumps the counter, the readers' poll sees it (1.4 us with
// the skew of 64 workgroups), and only THEN the gather starts its own round trip to memory.  But a frame's image is
// written exactly once per pass (one image per frame), so the scheduler fills the images with a pattern no h value
// can have (all ones: a bf16 NaN pair) before the pass, and the reader simply GATHERS - L2-bypassing loads - and
// looks at what it got: a 16-byte chunk with an all-ones dword is not there yet (16-byte stores may tear at dword
// granularity at worst, so every dword is checked), the wave sleeps and gathers again.  No drain, no counter, no poll
// on the dependency chain: publish -> visible in memory -> gathered.  The arrival counters remain for the SIDE
// streams (LayerNorm of the finished frames), bumped one step late behind a barrier that exists anyway.
template <bool TRACE, bool DP>
__global__ __launch_bounds__(256, 1) void stack_fwd_lpw_kernel(EdLpwLaunch L) {
    __shared__ float4 hand[4][3][4][64];      // cross-wave K sum: [source wave][slot of the owner][gate][lane], 48 KB
    if (ED_STEP_PRIO) __builtin_amdgcn_s_setprio(ED_STEP_PRIO);
    if (L.stamp && threadIdx.x == 0) atomicMin(&L.stamp[0], wall_clock64());
    const int B = L.B, H = L.H;
    const int UB = H >> 4, RG = (B + 63) >> 6, WGS = UB * RG;
    const int slot = blockIdx.x / WGS, rem = blockIdx.x - slot * WGS;
    const int ub = rem % UB, rg = rem / UB;
    const EdLpwSlot& S = L.slot[slot];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int KS = H >> 5, per = (KS + 3) >> 2;
    const int ks_beg = wave * per, ks_end = min(KS, ks_beg + per);
    const int MT = (B + 15) >> 4, mt0 = rg * 4, row0 = rg * 64;
    const long long H4 = 4ll * H, BH = (long long)B * H;

    // ---- stationary weights: this wave's K quarter of the slice, all 4 gates
    bf16x8_t w[LPW_PER][4];
    {
        const bf16_t* wbase = S.Wfrag + ((long long)ub * 4 * KS * 64 + lane) * 8;
#pragma unroll
        for (int i = 0; i < LPW_PER; ++i) {
            const int ks = min(ks_beg + i, KS - 1);          // clamp: duplicates are masked below
#pragma unroll
            for (int g = 0; g < 4; ++g) w[i][g] = ldfrag(wbase + ((long long)ks * 4 + g) * 512);
        }
    }
    if (S.wait_flag) soft_wait(S.wait_flag, L.err, 500u + slot);

    // ---- this lane's cells.  W is the FIRST MFMA operand below, so a lane ends up with 4 CONSECUTIVE units of one row
    // for each of the 4 gates: row 16 wave + (lane & 15) of the workgroup's 64 (wave w owns row tile w after the
    // cross-wave sum), units 4 q .. 4 q + 3 of its 16 (q = lane >> 4).  The cell update then needs no LDS at all:
    // pre-activations arrive as four 8-byte loads, the cell state stays in 4 registers for the whole launch, gates /
    // c / h leave as 8- and 16-byte stores (round 2's lane owned ONE unit of 4 rows: 44 two-byte LDS accesses per step)
    const int q = lane >> 4, r16 = lane & 15;
    const int brow = row0 + wave * 16 + r16;
    const bool live = brow < B;
    const long long goff = (long long)brow * H4 + ub * 64 + q * 4;       // + gate * 16
    const long long coff = (long long)brow * H + ub * 16 + q * 4;
    uint2 gpre[4];
    float4 cst = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        gpre[g] = make_uint2(0u, 0u);
        if (live) gpre[g] = *reinterpret_cast<const uint2*>(S.G + goff + g * 16);
    }
    if (live) cst = *reinterpret_cast<const float4*>(S.C_prev + coff);
    const __amdgpu_buffer_rsrc_t rimg = lpw_rsrc(S.img, (unsigned)S.img_bytes);
    __shared__ unsigned bail_s;
    if (tid == 0) bail_s = 0u;

    // debug trace (tools/lpw_trace.py): the first workgroup of every slot accumulates, per phase, the 100 MHz
    // ticks its lane 0 spent: [0] wait for the peers, [1] h loads + MFMA, [2] hand-off + cell update,
    // [3] publish (+ drain, !DP), [4] arrive (!DP) / gathers repeated x 100 (DP), [5] trailing stores; [6] steps, [7] launches
    const bool tr = TRACE && L.trace != nullptr && rem == 0 && tid == 0;
    long long ph[6] = {0, 0, 0, 0, 0, 0};
    long long last_ = tr ? wall_clock64() : 0;
#define LPW_STAMP(i)                                  \
    do {                                              \
        if (TRACE && tr) {                            \
            const long long now_ = wall_clock64();    \
            ph[i] += now_ - last_;                    \
            last_ = now_;                             \
        }                                             \
    } while (0)
    bool pend = false;                          // DP: the step before published, the side streams have not been told yet
    for (int s = 0; s < S.nsteps; ++s) {
        const int t = S.t0 + s;
        // ---- (1) !DP: every workgroup of this layer has published h_{t-1}
        if (!DP && s > 0 && tid == 0) {
            const unsigned want = S.base + (unsigned)(WGS * s);
            unsigned spins = 0;
            while (__hip_atomic_load(S.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 21)) {                  // ~ a second: a peer never became resident
                    if (L.err) atomicCAS(L.err, 0u, 700u + slot);
                    bail_s = 1u;
                    break;
                }
            }
        }
        // (a raw barrier: __syncthreads() would also wait for this wave's trailing stores and the prefetched
        // pre-activations - an HBM round trip that belongs behind the gather, not in front of it)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (bail_s) break;
        LPW_STAMP(0);
        // ---- (2) h_{t-1} fragments of this wave's K quarter: image t, plain loads (the
        // address has never been touched before in this pass - nothing stale anywhere - and the first reader on an
        // XCD brings a line into that XCD's L2 for the others)
        // Data polling: all fragments are requested, then looked at (a running maximum over their
        // dwords: all ones = not written yet), then multiplied.  A gather that came too early - rare: the gate / c stores
        // and the barrier above are delay enough (0.02 per step in the trace) - is simply done again.  (Using the
        // fragments as they land and reading the verdict off the accumulators - NaN in, NaN out - needs 400 registers or,
        // capped at the 368 that let a chunk product share the CU, spills into the cell update: 5.4 instead of 4.6 ms)
        bf16x8_t a[LPW_PER][4];
        {
            const unsigned soff_in = (unsigned)((long long)t * S.img_stride);
            unsigned tries = 0;
            for (;;) {
                unsigned worst = 0u;
#pragma unroll
                for (int i = 0; i < LPW_PER; ++i) {
                    const int ks = min(ks_beg + i, KS - 1);
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(
                            rimg, (unsigned)(((ks * MT + min(mt0 + m, MT - 1)) * 64 + lane) * 16), soff_in, DP ? 16 : 0);
                        a[i][m] = *reinterpret_cast<const bf16x8_t*>(&v);
                    }
                }
                if (!DP || s == 0) break;        // (step t0's image is complete since the launch before)
#pragma unroll
                for (int i = 0; i < LPW_PER; ++i)
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(&a[i][m]);
                        worst = max(worst, max(max(v[0], v[1]), max(v[2], v[3])));
                    }
                if (!__any(worst == 0xffffffffu)) break;
                if (TRACE && tr) ph[4] += 100;   // (the trace's "arrive" column then reads: gathers repeated per step)
                __builtin_amdgcn_s_sleep(2);
                if (++tries > (1u << 19)) {                  // ~ a second: a peer never became resident
                    if (L.err) atomicCAS(L.err, 0u, 700u + slot);
                    bail_s = 1u;                             // (every wave still goes to the barrier below)
                    break;
                }
            }
        }
        // ---- (3) W_hh h_{t-1}, two row tiles at a time (32 accumulator registers instead of 64: with the 128
        // weight registers the kernel has to stay under 368 so that a chunk-product workgroup of 144
        // registers per lane fits beside it on the CU), (4) partial tiles to their owners (wave m owns row
        // tile m) as soon as a pair is done
        f32x4_t mine[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) mine[g] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4_t acc[2][4];
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[mm][g] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < LPW_PER; ++i) {
                if (ks_beg + i < ks_end) {
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            acc[mm][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i][g], a[i][half * 2 + mm], acc[mm][g], 0, 0, 0);
                }
            }
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                const int m = half * 2 + mm;
                if (m != wave) {
                    const int sl = m < wave ? m : m - 1;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        hand[wave][sl][g][lane] = make_float4(acc[mm][g][0], acc[mm][g][1], acc[mm][g][2], acc[mm][g][3]);
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) mine[g] = acc[mm][g];
                }
            }
        }
        LPW_STAMP(1);
        // DP: every load of this step has landed in every wave, so the publish stores of the step before - older in each
        // wave's queue - are in memory: tell the side streams behind this barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (DP && bail_s) break;
        if (DP && pend && tid == 0) __hip_atomic_fetch_add(S.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pend = true;
        {
#pragma unroll
            for (int src = 0; src < 4; ++src) {
                if (src != wave) {
                    const int sl = wave < src ? wave : wave - 1;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 v = hand[src][sl][g][lane];
                        mine[g][0] += v.x; mine[g][1] += v.y; mine[g][2] += v.z; mine[g][3] += v.w;
                    }
                }
            }
            // ---- (5) cell update in registers: 4 units of one row, all four gates
            uint2 gout[4], hq;
            float4 cnew;
            {
                float hv[4], cv[4], o[4][4];
                const float cp[4] = {cst.x, cst.y, cst.z, cst.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    auto pre = [&](const uint2& u) {
                        const unsigned wd = (i < 2) ? u.x : u.y;
                        return __uint_as_float((i & 1) ? (wd & 0xffff0000u) : (wd << 16));
                    };
                    const float ig = fsigmoid(pre(gpre[0]) + mine[0][i]);
                    const float fg = fsigmoid(pre(gpre[1]) + mine[1][i]);
                    const float gg = ftanh(pre(gpre[2]) + mine[2][i]);
                    const float og = fsigmoid(pre(gpre[3]) + mine[3][i]);
                    const float c = fg * cp[i] + ig * gg;
                    hv[i] = og * ftanh(c);
                    cv[i] = c;
                    o[0][i] = ig; o[1][i] = fg; o[2][i] = gg; o[3][i] = og;
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    gout[g].x = f32x2_to_bf16x2(o[g][0], o[g][1]);
                    gout[g].y = f32x2_to_bf16x2(o[g][2], o[g][3]);
                }
                hq.x = f32x2_to_bf16x2(hv[0], hv[1]);
                hq.y = f32x2_to_bf16x2(hv[2], hv[3]);
                cnew = make_float4(cv[0], cv[1], cv[2], cv[3]);
            }
            LPW_STAMP(2);
            // ---- (6) publish FIRST: lanes l and l ^ 16 hold units 4q..4q+3 and 4q+4..4q+7 of one row - the even one stores
            // the 16-byte chunk of the next image and of the h row (the LayerNorm's input), write-through and BEFORE the
            // arrive (a side stream orders its LayerNorm launch behind this launch's steps by polling the counters)
            {
                const unsigned px = __shfl_xor(hq.x, 16), py = __shfl_xor(hq.y, 16);
                const int mt = mt0 + wave;
                if ((q & 1) == 0 && mt < MT) {
                    const u32x4_t vv = live ? (u32x4_t){hq.x, hq.y, px, py} : (u32x4_t){0u, 0u, 0u, 0u};
                    const int ks = ub >> 1, kg = (ub & 1) * 2 + (q >> 1);
                    __builtin_amdgcn_raw_buffer_store_b128(vv, rimg, (unsigned)((((ks * MT + mt) * 64) + kg * 16 + r16) * 16),
                                                           (unsigned)((long long)(t + 1) * S.img_stride), 16);
                    if (live) {
                        bf16_t* dst = S.Y + (long long)s * BH + (long long)brow * H + ub * 16 + (q >> 1) * 8;
                        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(vv) : "memory");
                    }
                }
            }
            if (!DP) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                LPW_STAMP(3);
                if (tid == 0) __hip_atomic_fetch_add(S.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                LPW_STAMP(4);
            } else {
                LPW_STAMP(3);
            }
            // ---- (7) off the chain: gates (for the backward pass), c rows; next step's pre-activations into registers
            cst = cnew;
            if (live) {
                bf16_t* G_t = S.G + (long long)s * B * H4;
#pragma unroll
                for (int g = 0; g < 4; ++g) *reinterpret_cast<uint2*>(G_t + goff + g * 16) = gout[g];
                *reinterpret_cast<float4*>(S.C + (long long)s * BH + coff) = cnew;
                if (s + 1 < S.nsteps) {
                    const bf16_t* G_n = G_t + (long long)B * H4;
#pragma unroll
                    for (int g = 0; g < 4; ++g) gpre[g] = *reinterpret_cast<const uint2*>(G_n + goff + g * 16);
                }
            }
            LPW_STAMP(5);
        }
    }
#undef LPW_STAMP
    if (DP && pend && !bail_s) {          // the last step's publish
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(S.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (TRACE && tr) {
        long long* o = L.trace + S.layer * 8;
        for (int i = 0; i < 6; ++i) atomicAdd((unsigned long long*)(o + i), (unsigned long long)ph[i]);
        atomicAdd((unsigned long long*)(o + 6), (unsigned long long)S.nsteps);
        atomicAdd((unsigned long long*)(o + 7), 1ull);
    }
    if (L.stamp) {
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(&L.stamp[1], wall_clock64());
    }
}


// LayerNorm of the frames [t0, t1) one layer has finished (weights-stationary path: the recurrence
// kernel carries whole chunks, so the norm of a chunk is its own small launch on the side stream):
// one workgroup per (output frame, 4 batch rows), same arithmetic as the norm role above.
struct ChunkNormArgs {
    const bf16_t* Yx1;      // h_t at Yx1 + t * B * H
    const bf16_t* X;        // residual rows (time-major) or null
    const float* gamma;
    const float* beta;
    bf16_t* out;            // frame tau, row b at out + tau * out_st + b * out_sb
    long long out_st, out_sb;
    float* mean;
    float* rstd;
    int B, H, T, t0, reduce;
    float eps;
    unsigned drop_thresh, drop_seed;
    float drop_scale;
    int drop_T;
};
__device__ __forceinline__ void chunk_norm_body(const ChunkNormArgs& a, int bid) {
    const int RB = (a.B + 3) >> 2;
    const int fo = bid / RB, rb = bid % RB;
    const long long BH = (long long)a.B * a.H;
    EdFwdNorm p;
    int ta, tb = -1;
    if (a.reduce == 1) ta = a.t0 + fo;
    else {
        ta = a.t0 + 2 * fo;                 // t0 is even under time reduction
        tb = ta + 1 < a.T ? ta + 1 : -1;    // an odd last frame has no partner: it counts as zero
    }
    const int tau = ta / a.reduce;
    p.y0 = a.Yx1 + (long long)ta * BH;
    p.r0 = a.X ? a.X + (long long)ta * BH : nullptr;
    p.y1 = tb >= 0 ? a.Yx1 + (long long)tb * BH : nullptr;
    p.r1 = (tb >= 0 && a.X) ? a.X + (long long)tb * BH : nullptr;
    p.gamma = a.gamma;
    p.beta = a.beta;
    p.out = a.out + (long long)tau * a.out_st;
    p.out_stride = a.out_sb;
    p.mean0 = a.mean + (long long)ta * a.B;
    p.rstd0 = a.rstd + (long long)ta * a.B;
    p.mean1 = tb >= 0 ? a.mean + (long long)tb * a.B : nullptr;
    p.rstd1 = tb >= 0 ? a.rstd + (long long)tb * a.B : nullptr;
    p.scale = a.reduce == 1 ? 1.f : 0.5f;
    p.drop_thresh = a.drop_thresh; p.drop_seed = a.drop_seed; p.drop_scale = a.drop_scale; p.drop_T = a.drop_T;
    p.tau = tau;
    fwd_norm_role(p, rb, a.B, a.H, a.eps);
}

// the frames several layers finished in ONE launch-persistent launch, normalised by one launch (the side
// stream then carries one norm launch per recurrence launch instead of one per layer)
struct MultiNormArgs {
    ChunkNormArgs a[ED_STACK_MAX_SLOTS];
    int first[ED_STACK_MAX_SLOTS + 1];     // first workgroup of item i
    int n;
};
__global__ __launch_bounds__(256) void stack_multi_norm_kernel(MultiNormArgs M) {
    int i = 0;
    while (i + 1 < M.n && (int)blockIdx.x >= M.first[i + 1]) ++i;
    chunk_norm_body(M.a[i], (int)blockIdx.x - M.first[i]);
}

// =====================================================================================
// backward (BPTT step)
// =====================================================================================
#ifndef ED_BCH
#define ED_BCH 4
#endif
#ifndef ED_BWD_OCC
#define ED_BWD_OCC 2
#endif
constexpr int BCH = ED_BCH;   // k-steps per register buffer: 4 x (2 A + 2 W) x 16 B = 256 B / lane, two buffers

struct __attribute__((aligned(16))) BwdShared {
    float4 hand[4][3][64];   // [source wave][destination slot][lane]   12 KB
    bf16_t g[32][136];       // gates in, dG out (128 cols + 8 pad)
    float ct[32][36];        // c_t
    float cp[32][36];        // c_{t-1}
    float dc[32][36];        // dL/dc running, in/out
    bf16_t dy[32][40];       // dL/dh_t from above
};

__device__ __forceinline__ void bwd_step_role(const EdBwdStep& p, int nb, int rg, int B, int H,
                                              BwdShared& sh) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int KS = H >> 3;   // 4H / 32
    const int per = (KS + 3) >> 2;
    const int ks_beg = wave * per, ks_end = min(KS, ks_beg + per);
    const int MT = (B + 15) >> 4, mt0 = rg * 2, row0 = rg * 32;
    const long long H4 = 4ll * H;

    uint4 gin[2], dyin = make_uint4(0, 0, 0, 0);
    float4 ctin, cpin, dcin;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + 256 * i, r = id >> 4, part = id & 15, b = row0 + r;
        gin[i] = make_uint4(0, 0, 0, 0);
        if (b < B) gin[i] = *reinterpret_cast<const uint4*>(p.G_t + b * H4 + nb * 128 + part * 8);
    }
    {
        const int r = tid >> 3, part = tid & 7, b = row0 + r;
        ctin = cpin = dcin = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < B) {
            const long long o = (long long)b * H + nb * 32 + part * 4;
            ctin = *reinterpret_cast<const float4*>(p.C_t + o);
            cpin = *reinterpret_cast<const float4*>(p.C_prev + o);
            dcin = *reinterpret_cast<const float4*>(p.dC + o);
        }
        if (tid < 128) {
            const int r2 = tid >> 2, part2 = tid & 3, b2 = row0 + r2;
            if (p.dY_t && b2 < B)
                dyin = *reinterpret_cast<const uint4*>(p.dY_t + (long long)b2 * H + nb * 32 + part2 * 8);
        }
    }

    f32x4_t acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (p.gfrag_in) {
        const bf16_t* abase = p.gfrag_in + lane * 8;
        const bf16_t* wbase = p.WTfrag + ((long long)nb * 2 * KS * 64 + lane) * 8;
        bf16x8_t a0[BCH][2], w0[BCH][2], a1[BCH][2], w1[BCH][2];
        auto load = [&](bf16x8_t (&a)[BCH][2], bf16x8_t (&w)[BCH][2], int ks0) {
#pragma unroll
            for (int i = 0; i < BCH; ++i) {
                const int ks = min(ks0 + i, ks_end - 1);
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    // unconditional (a duplicate tile past the batch): a conditional load is an exec-masked branch
                    // per fragment, and across those blocks the compiler's wait counts collapsed to vmcnt(6..7) -
                    // half a buffer in flight instead of two
                    a[i][m] = ldfrag(abase + ((long long)ks * MT + min(mt0 + m, MT - 1)) * 512);
#pragma unroll
                for (int n = 0; n < 2; ++n) w[i][n] = ldfrag(wbase + ((long long)ks * 2 + n) * 512);
            }
        };
        auto mma = [&](bf16x8_t (&a)[BCH][2], bf16x8_t (&w)[BCH][2], int ks0) {
#pragma unroll
            for (int i = 0; i < BCH; ++i) {
                if (ks0 + i < ks_end) {
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][m], w[i][n], acc[m][n], 0, 0, 0);
                }
            }
        };
        if (ks_beg < ks_end) {       // no conditional loads in the steady state: see fwd_step_role
            load(a0, w0, ks_beg);
            int ks0 = ks_beg;
            for (; ks0 + 2 * BCH < ks_end; ks0 += 2 * BCH) {
                load(a1, w1, ks0 + BCH);
                __builtin_amdgcn_sched_barrier(0);
                mma(a0, w0, ks0);
                __builtin_amdgcn_sched_barrier(0);
                load(a0, w0, ks0 + 2 * BCH);
                __builtin_amdgcn_sched_barrier(0);
                mma(a1, w1, ks0 + BCH);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ks0 + BCH < ks_end) {
                load(a1, w1, ks0 + BCH);
                mma(a0, w0, ks0);
                mma(a1, w1, ks0 + BCH);
            } else {
                mma(a0, w0, ks0);
            }
        }
    }

    // wave w owns tile (m = w >> 1, n = w & 1)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        if (tt != wave) {
            const int slot = tt < wave ? tt : tt - 1;
            const f32x4_t v = acc[tt >> 1][tt & 1];
            sh.hand[wave][slot][lane] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + 256 * i, r = id >> 4, part = id & 15;
        *reinterpret_cast<uint4*>(&sh.g[r][part * 8]) = gin[i];
    }
    {
        const int r = tid >> 3, part = tid & 7;
        *reinterpret_cast<float4*>(&sh.ct[r][part * 4]) = ctin;
        *reinterpret_cast<float4*>(&sh.cp[r][part * 4]) = cpin;
        *reinterpret_cast<float4*>(&sh.dc[r][part * 4]) = dcin;
        if (tid < 128) *reinterpret_cast<uint4*>(&sh.dy[tid >> 2][(tid & 3) * 8]) = dyin;
    }
    __syncthreads();

    f32x4_t mine = wave == 0 ? acc[0][0] : wave == 1 ? acc[0][1] : wave == 2 ? acc[1][0] : acc[1][1];
#pragma unroll
    for (int src = 0; src < 4; ++src) {
        if (src != wave) {
            const int slot = wave < src ? wave : wave - 1;
            const float4 v = sh.hand[src][slot][lane];
            mine[0] += v.x; mine[1] += v.y; mine[2] += v.z; mine[3] += v.w;
        }
    }

    const int m = wave >> 1, n = wave & 1;
    const int u = lane & 15, ul = n * 16 + u, cb = n * 64 + u, rbase = m * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rl = rbase + q;
        const float ig = bf16_to_f32(sh.g[rl][cb]), fg = bf16_to_f32(sh.g[rl][cb + 16]);
        const float gg = bf16_to_f32(sh.g[rl][cb + 32]), og = bf16_to_f32(sh.g[rl][cb + 48]);
        const float dh = bf16_to_f32(sh.dy[rl][ul]) + mine[q];
        const float tc = ftanh(sh.ct[rl][ul]);
        const float dct = sh.dc[rl][ul] + dh * og * (1.f - tc * tc);
        sh.g[rl][cb] = f32_to_bf16(dct * gg * ig * (1.f - ig));
        sh.g[rl][cb + 16] = f32_to_bf16(dct * sh.cp[rl][ul] * fg * (1.f - fg));
        sh.g[rl][cb + 32] = f32_to_bf16(dct * ig * (1.f - gg * gg));
        sh.g[rl][cb + 48] = f32_to_bf16(dh * tc * og * (1.f - og));
        sh.dc[rl][ul] = dct * fg;
    }
    __syncthreads();

    // 1280 16-byte stores: dG plain 512, dG fragment image 512, dC 256
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int task = tid + 256 * i;
        if (task < 512) {
            const int r = task >> 4, part = task & 15, b = row0 + r;
            if (b < B)
                *reinterpret_cast<uint4*>(p.G_t + b * H4 + nb * 128 + part * 8) =
                    *reinterpret_cast<const uint4*>(&sh.g[r][part * 8]);
        } else if (task < 1024) {
            const int id = task - 512, mm = id >> 8, ksl = (id >> 6) & 3, kg = (id >> 4) & 3, r16 = id & 15;
            const int mt = mt0 + mm;
            if (mt < MT && p.gfrag_out) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (mt * 16 + r16 < B)
                    v = *reinterpret_cast<const uint4*>(&sh.g[mm * 16 + r16][ksl * 32 + kg * 8]);
                *reinterpret_cast<uint4*>(p.gfrag_out + (((long long)(nb * 4 + ksl) * MT + mt) * 64 + kg * 16 + r16) * 8) = v;
            }
        } else {
            const int id = task - 1024, r = id >> 3, part = id & 7, b = row0 + r;
            if (b < B)
                *reinterpret_cast<float4*>(p.dC + (long long)b * H + nb * 32 + part * 4) =
                    *reinterpret_cast<const float4*>(&sh.dc[r][part * 4]);
        }
    }
}

__global__ __launch_bounds__(256, ED_BWD_OCC) void stack_bwd_kernel(EdBwdLaunch L) {
    __shared__ BwdShared sh;
    if (ED_STEP_PRIO) __builtin_amdgcn_s_setprio(ED_STEP_PRIO);
    if (L.stamp && threadIdx.x == 0) atomicMin(&L.stamp[0], wall_clock64());   // see stack_fwd_kernel
    const int NB = L.H >> 5, RG = (L.B + 31) >> 5;
    const int bid = blockIdx.x;
    const int slot = bid / (NB * RG), rem = bid - slot * NB * RG;
    if (L.step[slot].wait_flag) soft_wait(L.step[slot].wait_flag, L.err, 600u + slot);
    bwd_step_role(L.step[slot], rem % NB, rem / NB, L.B, L.H, sh);
    if (L.stamp) {
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(&L.stamp[1], wall_clock64());
    }
}

// =====================================================================================
// backward, split-K + weights-stationary (EdSkLaunch, stack_kernels.hpp).  Workgroup (unit block ub of 64 units,
// K quarter kq of the 4H interleaved gate columns), all 64 rows:
//   partial[row][unit] = sum_{k in quarter} dG_{t+1}[row][k] W_hh[row(k)][unit]
//       wave n owns unit tile n (16 units) for ALL rows and the WHOLE quarter: its 32 W fragments stay in 128
//       registers for the launch; the dG fragments (128 KB per step, the only dependent fetch) go through a ring of
//       three 16 KB LDS slots by LDS-DMA (wave w brings row tile w of every k-step) with counted waits and raw
//       barriers, every wave reads all of them: no cross-wave sum, no register buffers for the stream;
//       W is the FIRST MFMA operand, so a lane holds 4 CONSECUTIVE units of one row - 16-byte partial stores,
//       8-byte gate accesses in the cell update;
//   the 4 workgroups of a unit block publish their partials (write-through), meet on the block's counter, and
//   workgroup kq finishes rows [16 kq, 16 kq + 16): dh = sum of the 4 partials in fixed order, cell backward in
//   registers (dL/dc never leaves them inside a launch), dG_t into image t (write-through) and into G (plain).
// Blocks of one launch slot are numbered ub * 4 + kq: blockIdx % 8 = 4 (ub & 1) + kq, so an XCD only ever reads
// ONE quarter of a layer's dG image (128 KB, shared by its 8 workgroups through L2).
// Registers <= 256 and LDS 48 KB: a gemm_tn256 workgroup (2 waves per SIMD x 128 registers, 96 KB) fits beside it.
// =====================================================================================
#ifndef ED_SK_GK
#define ED_SK_GK 4
#endif
#ifndef ED_SK_RING
#define ED_SK_RING 3
#endif
#ifndef ED_SK_ABLATE
#define ED_SK_ABLATE 0      // timing experiments only (results wrong): 1 no dG stream, 2 no LDS reads / MFMA
#endif
#ifndef ED_SK_TAIL
#define ED_SK_TAIL 0        // timing experiments only (results wrong): 1 no row-form dG stores, 2 the cell operands of the
#endif                      // first step for every step (no re-request).  profiles/r6_sk_tail.txt: BOTH make the pass slower
constexpr int SK_GK = ED_SK_GK;             // k-steps per ring slot
constexpr int SK_SLOT = SK_GK * 4 * 1024;   // 16 KB: 4 k-steps x 4 row tiles x 1 KB
constexpr int SK_MAXG = 32 / SK_GK;         // ring slots per step at H = 1024
constexpr int SK_RING = ED_SK_RING;         // slots: RING - 1 of them in flight while one is consumed
// s_waitcnt vmcnt(n * SK_GK): all but the n youngest slots of this wave's DMA have landed
template <int N>
__device__ __forceinline__ void sk_wait_slots() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N * SK_GK) : "memory");
}
__device__ __forceinline__ void sk_wait_younger(int n) {
    if (n <= 0) sk_wait_slots<0>();
    else if (n == 1) sk_wait_slots<1>();
    else if (n == 2) sk_wait_slots<2>();
    else if (n == 3) sk_wait_slots<3>();
    else if (n == 4) sk_wait_slots<4>();
    else if (n == 5) sk_wait_slots<5>();
    else sk_wait_slots<6>();
}
static_assert(SK_RING >= 2 && SK_RING <= 8 && (SK_RING - 2) * SK_GK < 64, "ring depth");
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

// (Round 4, measured and removed: (a) the workgroup's rows as 2 / 4 sub-batches with their own counters, the arrival of
// a partial deferred into the next product and the finishing block one product later - bit-identical, 9.0 -> 9.2 / 11.3 ms
// per backward pass: a product of half the rows takes 60 % of the time of the full one and the deferred arrival comes a
// whole product late; (b) data polling as in the forward kernel - dG images and a ring of partial buffers pre-filled with
// NaN, every wave reading its own four LDS-DMA pieces back before the slot barrier: the ring phase went from 5.6 to
// 9.2 us per step, the read-back serialises what the counted waits had pipelined.)
template <bool TRACE>
__global__ __launch_bounds__(256, 2) void stack_bwd_sk_kernel(EdSkLaunch L) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];      // SK_RING * SK_SLOT bytes (dynamic)
    __shared__ unsigned bail_s;
    if (ED_STEP_PRIO) __builtin_amdgcn_s_setprio(ED_STEP_PRIO);
    if (L.stamp && threadIdx.x == 0) atomicMin(&L.stamp[0], wall_clock64());
    const int B = L.B, H = L.H;
    const int UBK = H >> 6, WGS = UBK * 4;
    const int slot = blockIdx.x / WGS, rem = blockIdx.x - slot * WGS;
    const int ub = rem >> 2, kq = rem & 3;
    const EdSkSlot& S = L.slot[slot];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int KSq = H >> 5;                         // k-steps of this workgroup's quarter (<= 32)
    const int NGRP = (KSq + SK_GK - 1) / SK_GK;     // ring slots per step (<= SK_MAXG)
    // the 8 workgroups of an XCD that read the same K quarter (same kq, same parity of ub) start their walk over the
    // quarter's k-groups at 8 different places: the first window of each brings a different eighth of the image
    // into the XCD's L2, the later windows of all of them hit it (fresh write-through data is an L2 miss for the
    // first reader; in lock-step every workgroup paid that miss on every window)
    const int rot = (ub >> 1) % NGRP;
    const int MT = (B + 15) >> 4;
    const long long H4 = 4ll * H, BH = (long long)B * H;
    const __amdgpu_buffer_rsrc_t rimg = lpw_rsrc(S.img, (unsigned)S.img_bytes);
    const __amdgpu_buffer_rsrc_t rpart = lpw_rsrc(S.part, (unsigned)(2u * UBK * 4u * 64u * 64u * 4u));
    // who this workgroup waits for: quarter kq of the gate columns is the 4 gates of units
    // [kq H/4, (kq+1) H/4), i.e. of unit blocks [kq UBK/4, (kq+1) UBK/4) - 16 of the layer's 64 workgroups at H = 1024.
    // One arrival counter per quarter (a 256-byte line each, behind the unit-block counters) when the blocks do not
    // straddle quarters; otherwise the layer's single counter
    const bool quarters = (UBK & 3) == 0;
    unsigned* cnt_arrive = quarters ? S.gcounter + (SK_CNT_QUARTER + ub / (UBK >> 2)) * 64 : S.counter;
    const unsigned* cnt_poll = quarters ? S.gcounter + (SK_CNT_QUARTER + kq) * 64 : S.counter;

    // ---- stationary weights: unit tile `wave` of the (ub, kq) slice, every k-step of the quarter
    bf16x8_t w[32];
    {
        const bf16_t* wbase = S.Wsk + (((((long long)(ub * 4 + kq) * KSq) * 4) + wave) * 64 + lane) * 8;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int ks = (min(i, NGRP * SK_GK - 1) + rot * SK_GK) % (NGRP * SK_GK);      // register i <-> k-step, rotated
            w[i] = ldfrag(wbase + (long long)min(ks, KSq - 1) * 4 * 512);
        }
    }
    if (S.wait_flag) soft_wait(S.wait_flag, L.err, 600u + slot);

    // ---- this lane's cells: row 16 kq + (lane & 15), units ub*64 + quad*4 .. +3
    const int crow = kq * 16 + (lane & 15), quad = wave * 4 + (lane >> 4);
    const bool live = crow < B;
    const int unit0 = ub * 64 + quad * 4;
    const int gcol0 = (unit0 >> 4) * 64 + (unit0 & 15);      // + gate * 16
    float4 dc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) dc = *reinterpret_cast<const float4*>(S.dC + (long long)crow * H + unit0);
    // partial tile position of this lane in the MFMA result: rows 16 m + (lane & 15), units wave*16 + (lane >> 4)*4 ..
    const unsigned pmine = (unsigned)(((lane & 15) * 64 + wave * 16 + (lane >> 4) * 4) * 4);
    if (tid == 0) bail_s = 0u;

    // debug trace (tools/sk_trace.py): workgroup 0 of every slot accumulates the 100 MHz ticks its lane 0 spent in
    // [0] wait for the quarter's producers, [1] dG ring + MFMA + partial stores, [2] drain + unit-block wait + partial
    // reads, [3] cell, [4] publish + drain + arrive, [5] trailing stores; [6] steps, [7] launches
    const bool tr = TRACE && L.trace != nullptr && rem == 0 && tid == 0;
    long long ph[6] = {0, 0, 0, 0, 0, 0};
    long long last_ = tr ? wall_clock64() : 0;
#define SK_STAMP(i)                                   \
    do {                                              \
        if (TRACE && tr) {                            \
            const long long now_ = wall_clock64();    \
            ph[i] += now_ - last_;                    \
            last_ = now_;                             \
        }                                             \
    } while (0)
    // ---- a frame's cell operands (HBM: the forward pass wrote them long ago).  Requested at the END of the step before
    // (and here for the first one): the wait for the peers is their flight time, and nothing of this wave is outstanding
    // in front of the first ring slot except loads that are about to land
    uint2 gq[4], dyq = make_uint2(0u, 0u);
    float4 ctq = make_float4(0.f, 0.f, 0.f, 0.f), cpq = ctq;
#pragma unroll
    for (int g = 0; g < 4; ++g) gq[g] = make_uint2(0u, 0u);
    auto request_operands = [&](int s) {
        if (live) {
            const int t = S.t0 - s;
            const bf16_t* G_t = S.G - (long long)s * B * H4 + (long long)crow * H4 + gcol0;
#pragma unroll
            for (int g = 0; g < 4; ++g) gq[g] = *reinterpret_cast<const uint2*>(G_t + g * 16);
            const long long o = (long long)crow * H + unit0;
            ctq = *reinterpret_cast<const float4*>(S.Cx + (long long)(t + 1) * BH + o);
            cpq = *reinterpret_cast<const float4*>(S.Cx + (long long)t * BH + o);
            if (S.dY) dyq = *reinterpret_cast<const uint2*>(S.dY - (long long)s * BH + o);
        }
    };
    request_operands(0);
    for (int s = 0; s < S.nsteps; ++s) {
        const int t = S.t0 - s;
        // ---- (1) the producers of this quarter have published dG_{t+1}
        if (tid == 0) {
            unsigned bail = 0u;
            const unsigned want = (quarters ? (unsigned)UBK : (unsigned)WGS) * (S.done + (unsigned)s);
            unsigned spins = 0;
            while (s > 0 && __hip_atomic_load(cnt_poll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 21)) {
                    if (L.err) atomicCAS(L.err, 0u, 900u + slot);
                    bail = 1u;
                    break;
                }
            }
            bail_s = bail;
        }
        // (a raw barrier: __syncthreads() would make every wave wait for its own operand loads here)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (bail_s) break;
        SK_STAMP(0);
        const int pr = t & 1;
        const unsigned pbase = (unsigned)((((pr * UBK + ub) * 4 + kq) * 64 * 64) * 4);
        f32x4_t acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (t < S.T - 1) {
            // ---- (2) partial product of this K quarter.  Ring slot = 4 k-steps x 4 row tiles; wave w brings row tile
            // w (clamped to the batch) of each k-step: one 1 KB LDS-DMA per (wave, k-step)
            const unsigned char* img = reinterpret_cast<const unsigned char*>(S.img) + (long long)(t + 1) * S.img_stride;
            const int mrow = min(wave, MT - 1);
            const unsigned avoff = (unsigned)(((kq * KSq * MT + mrow) * 64 + lane) * 16);     // + scalar base: one VGPR
            auto issue = [&](int g) {
                unsigned char* dst = ring + (g % SK_RING) * SK_SLOT + wave * 1024;
#pragma unroll
                for (int j = 0; j < SK_GK; ++j) {
                    int gr = g + rot;
                    if (gr >= NGRP) gr -= NGRP;
                    const int ks = min(gr * SK_GK + j, KSq - 1);
                    // hand-written: the compiler makes every LDS read wait for ALL outstanding LDS-DMA it knows of
                    // (vmcnt(0) right after the next slot's issue); the counted waits below are the real rule
                    const unsigned m0v = __builtin_amdgcn_readfirstlane(
                        (unsigned)(size_t)(__attribute__((address_space(3))) void*)(dst + j * 4096));
                    // m0 is saved and restored INSIDE the statement: naming a reserved register as a clobber is undefined
                    // behaviour for the register allocator (the instruction reads m0 at issue, so the restore may follow
                    // it at once)
                    unsigned m0_keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                                 "s_mov_b32 m0, %0"
                                 : "=&s"(m0_keep)
                                 : "v"(avoff), "s"(m0v), "s"(img + (long long)ks * MT * 1024) : "memory");
                }
            };
            // (older loads of this wave - the cell operands - only make the counted waits below conservative)
#pragma unroll
            for (int g = 0; g < SK_RING - 1; ++g)
                if (ED_SK_ABLATE != 1 && g < NGRP) issue(g);
#pragma unroll
            for (int g = 0; g < SK_MAXG; ++g) {
                if (g < NGRP) {
                    sk_wait_younger(min(SK_RING - 2, NGRP - 1 - g));
                    asm volatile("s_barrier" ::: "memory");
                    if (ED_SK_ABLATE != 1 && g + SK_RING - 1 < NGRP) issue(g + SK_RING - 1);
                    if (ED_SK_ABLATE == 2) continue;
                    const unsigned char* src = ring + (g % SK_RING) * SK_SLOT + lane * 16;
                    bf16x8_t af[2][4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) af[0][m] = *reinterpret_cast<const bf16x8_t*>(src + m * 1024);
#pragma unroll
                    for (int j = 0; j < SK_GK; ++j) {
                        if (j + 1 < SK_GK) {
#pragma unroll
                            for (int m = 0; m < 4; ++m)
                                af[(j + 1) & 1][m] = *reinterpret_cast<const bf16x8_t*>(src + (j + 1) * 4096 + m * 1024);
                        }
                        if (KSq % SK_GK == 0 || ((g + rot) % NGRP) * SK_GK + j < KSq) {
#pragma unroll
                            for (int m = 0; m < 4; ++m)
                                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[g * SK_GK + j], af[j & 1][m], acc[m], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        // ---- (3) publish the partial (zeros at the last frame: it has no dG_{t+1}; the exchange still runs - one rule
        // for every step): rows 16 m + (lane & 15), 4 consecutive units per lane, write-through
        // Row tile kq is finished by THIS workgroup, and the MFMA layout already puts the partial of this lane's own cells
        // (row 16 kq + (lane & 15), units quad*4 ..) into acc[kq] of this very lane: it stays in registers - three
        // write-through stores per wave to drain instead of four (~0.43 us each, one behind the other)
        f32x4_t own = acc[0];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m == kq) {
                own = acc[m];
            } else {
                const u32x4_t o = {__float_as_uint(acc[m][0]), __float_as_uint(acc[m][1]), __float_as_uint(acc[m][2]), __float_as_uint(acc[m][3])};
                __builtin_amdgcn_raw_buffer_store_b128(o, rpart, pbase + pmine + (unsigned)(m * 16 * 64 * 4), 0, 16);
            }
        }
        SK_STAMP(1);
        // ---- (4) dL/dh_t of this lane's cells: the four K parts in fixed order
        float dh[4] = {0.f, 0.f, 0.f, 0.f};
        {
            // every storing wave drains; one lane arrives on the unit block's counter and waits for the other three
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                unsigned* gc = S.gcounter + ub * 64;      // one 256-byte line per unit block
                __hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned want = 4u * (S.done + (unsigned)s + 1u);
                unsigned spins = 0, bail = 0u;
                while (__hip_atomic_load(gc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 21)) {
                        if (L.err) atomicCAS(L.err, 0u, 950u + slot);
                        bail = 1u;
                        break;
                    }
                }
                bail_s = bail;
            }
            __syncthreads();
            if (bail_s) break;
            // (the same order of the four K parts for every row, whoever finishes it: q = 0, 1, 2, 3, the own one from
            // its registers)
            u32x4_t pv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                pv[q] = (u32x4_t){0u, 0u, 0u, 0u};
                if (q != kq)
                    pv[q] = __builtin_amdgcn_raw_buffer_load_b128(
                        rpart, (unsigned)((((((pr * UBK + ub) * 4 + q) * 64 + crow) * 64) + quad * 4) * 4), 0, 16);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool mine = q == kq;
#pragma unroll
                for (int e = 0; e < 4; ++e) dh[e] += mine ? own[e] : __uint_as_float(pv[q][e]);
            }
        }
        SK_STAMP(2);
        // ---- (5) cell backward in registers
        uint2 outg[4];
        {
            const float ctv[4] = {ctq.x, ctq.y, ctq.z, ctq.w}, cpv[4] = {cpq.x, cpq.y, cpq.z, cpq.w};
            float dcv[4] = {dc.x, dc.y, dc.z, dc.w};
            float o[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                auto pick = [&](const uint2& u) {
                    const unsigned wd = (j < 2) ? u.x : u.y;
                    return __uint_as_float((j & 1) ? (wd & 0xffff0000u) : (wd << 16));
                };
                const float ig = pick(gq[0]), fg = pick(gq[1]), gg = pick(gq[2]), og = pick(gq[3]);
                const float dht = pick(dyq) + dh[j];
                const float tc = ftanh(ctv[j]);
                const float dct = dcv[j] + dht * og * (1.f - tc * tc);
                o[0][j] = dct * gg * ig * (1.f - ig);
                o[1][j] = dct * cpv[j] * fg * (1.f - fg);
                o[2][j] = dct * ig * (1.f - gg * gg);
                o[3][j] = dht * tc * og * (1.f - og);
                dcv[j] = dct * fg;
            }
            dc = make_float4(dcv[0], dcv[1], dcv[2], dcv[3]);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                outg[g].x = f32x2_to_bf16x2(o[g][0], o[g][1]);
                outg[g].y = f32x2_to_bf16x2(o[g][2], o[g][3]);
            }
        }
        SK_STAMP(3);
        // ---- (6) publish dG_t into image t.  Lanes l and l ^ 16 hold units u..u+3 and u+4..u+7 of one row: they swap
        // two gates each, so every lane holds two whole 16-byte chunks (8 consecutive gate columns), stored write-through
        u32x4_t v16[2];
        const bool hi = (lane >> 4) & 1;
        const int kc0 = (gcol0 & ~7) + (hi ? 32 : 0);        // first of the pair's 8 columns, gate 2 (hi) or 0
        {
            // (bit masks, not ?: - the compiler turns a select between two array elements into an indexed scratch load)
            const unsigned himask = 0u - (unsigned)hi;
            const unsigned g0x = (outg[0].x & himask) | (outg[2].x & ~himask), g0y = (outg[0].y & himask) | (outg[2].y & ~himask);
            const unsigned g1x = (outg[1].x & himask) | (outg[3].x & ~himask), g1y = (outg[1].y & himask) | (outg[3].y & ~himask);
            uint2 got[2];
            got[0].x = __shfl_xor(g0x, 16); got[0].y = __shfl_xor(g0y, 16);
            got[1].x = __shfl_xor(g1x, 16); got[1].y = __shfl_xor(g1y, 16);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                uint2 own;
                own.x = (outg[2 + i].x & himask) | (outg[i].x & ~himask);
                own.y = (outg[2 + i].y & himask) | (outg[i].y & ~himask);
                v16[i] = hi ? (u32x4_t){got[i].x, got[i].y, own.x, own.y} : (u32x4_t){own.x, own.y, got[i].x, got[i].y};
            }
            if (t > 0 && live) {
                const unsigned soff_out = (unsigned)((long long)t * S.img_stride);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int kc = kc0 + i * 16;
                    const int ks = kc >> 5, kg = (kc >> 3) & 3;
                    __builtin_amdgcn_raw_buffer_store_b128(
                        v16[i], rimg, (unsigned)(((ks * MT + (crow >> 4)) * 64 + kg * 16 + (crow & 15)) * 16), soff_out, 16);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(cnt_arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        SK_STAMP(4);
        // ---- (7) off the chain: the dG rows the dX / weight-gradient products read (later kernels, ordered by an event
        // behind this launch), the same two chunks; then the next frame's cell operands
        if (live && !(ED_SK_TAIL & 1)) {
            bf16_t* G_t = S.G - (long long)s * B * H4 + (long long)crow * H4 + kc0;
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4_t*>(G_t + i * 16) = v16[i];
        }
        if (s + 1 < S.nsteps && !(ED_SK_TAIL & 2)) request_operands(s + 1);
        SK_STAMP(5);
    }
#undef SK_STAMP
    if (TRACE && tr) {
        long long* o = L.trace + S.layer * 8;
        for (int i = 0; i < 6; ++i) atomicAdd((unsigned long long*)(o + i), (unsigned long long)ph[i]);
        atomicAdd((unsigned long long*)(o + 6), (unsigned long long)S.nsteps);
        atomicAdd((unsigned long long*)(o + 7), 1ull);
    }
    if (live) *reinterpret_cast<float4*>(S.dC + (long long)crow * H + unit0) = dc;
    if (L.stamp) {
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(&L.stamp[1], wall_clock64());
    }
}

// split-K fragment image of W_hh: frag[ub H/64][kq 4][ksl H/32][n 4][lane 64][8], W as the FIRST MFMA operand:
// row index lane & 15 -> unit ub*64 + n*16 + (lane & 15); k = interleaved gate column kq*H + ksl*32 + (lane >> 4)*8 + e
// thread = one 16-byte chunk: 8 consecutive gate columns = 8 consecutive W rows at one unit; the four waves of a block
// cover 64 consecutive units (256 contiguous bytes of each row)
__global__ void pack_whh_sk_kernel(const float* __restrict__ W, bf16_t* __restrict__ out, int H) {
    const long long n8 = 4ll * H * H / 8;
    const int KSq = H >> 5;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < n8; c += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(c & 63), nn = (int)((c >> 6) & 3);
        const long long blk = c >> 8;
        const int ksl = (int)(blk % KSq), kq = (int)((blk / KSq) & 3), ub = (int)(blk / KSq / 4);
        const int unit = ub * 64 + nn * 16 + (lane & 15);
        const int kcol = kq * H + ksl * 32 + (lane >> 4) * 8;
        const int ubk = kcol >> 6, g = (kcol >> 4) & 3, u = kcol & 15;
        const float* src = W + ((long long)g * H + ubk * 16 + u) * H + unit;
        unsigned o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o[e] = (unsigned)f32_to_bf16(src[(long long)(2 * e) * H]) | ((unsigned)f32_to_bf16(src[(long long)(2 * e + 1) * H]) << 16);
        *reinterpret_cast<uint4*>(out + c * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// =====================================================================================
// LayerNorm backward, time-major, frames [t0, t1) of one layer.  One wave per input row (t, b):
//   z = y (+ res), xhat = (z - mean) rstd, dy = dout[t / reduce, b] / reduce, g = dy gamma
//   dz = rstd (g - mean_H(g) - xhat mean_H(g xhat));  dgamma += dy xhat;  dbeta += dy
// Lane owns columns lane*8 + 512 i (i < 4, H <= 2048): one pass over the row in registers,
// dgamma / dbeta accumulated per lane over the wave's rows, combined through LDS into ONE partial
// row per workgroup (no atomics: 2H x grid contended fp32 atomics per launch cost 20 us).
// =====================================================================================
constexpr int LNB_NB = 4;
__global__ __launch_bounds__(256) void stack_ln_bwd_kernel(
    const bf16_t* __restrict__ dout, long long dout_st, long long dout_sb,
    const bf16_t* __restrict__ y, const bf16_t* __restrict__ res, const float* __restrict__ gamma,
    const float* __restrict__ mean_in, const float* __restrict__ rstd_in, bf16_t* __restrict__ dz,
    float* __restrict__ part, int B, int H, int t0, int t1, int reduce, unsigned drop_thresh, unsigned drop_seed,
    float drop_scale, int drop_T) {
    __shared__ float red[2][3][LNB_NB * 8][64];   // waves 1..3 -> wave 0, 48 KB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float scale = 1.f / (float)reduce;
    float ag[LNB_NB][8], ab[LNB_NB][8], gm[LNB_NB][8];
#pragma unroll
    for (int i = 0; i < LNB_NB; ++i) {
        const int c = lane * 8 + 512 * i;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ag[i][e] = 0.f;
            ab[i][e] = 0.f;
            gm[i][e] = (c < H) ? gamma[c + e] : 0.f;
        }
    }
    const long long rows = (long long)(t1 - t0) * B;
    for (long long rr = (long long)blockIdx.x * 4 + wave; rr < rows; rr += (long long)gridDim.x * 4) {
        const int t = t0 + (int)(rr / B), b = (int)(rr % B);
        const long long row = (long long)t * B + b;
        const bf16_t* dyr = dout + (long long)(t / reduce) * dout_st + (long long)b * dout_sb;
        const bf16_t* yr = y + row * H;
        const bf16_t* rsr = res ? res + row * H : nullptr;
        const float m = mean_in[row], rs = rstd_in[row];
        float dy[LNB_NB][8], xh[LNB_NB][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LNB_NB; ++i) {
            const int c = lane * 8 + 512 * i;
            if (c < H) {
                float z[8];
                ElemIO<bf16_t>::load_vec(dyr + c, dy[i]);
                ElemIO<bf16_t>::load_vec(yr + c, z);
                if (rsr) {
                    float r2[8];
                    ElemIO<bf16_t>::load_vec(rsr + c, r2);
#pragma unroll
                    for (int e = 0; e < 8; ++e) z[e] += r2[e];
                }
                if (drop_thresh) {
                    // the gradient arrives at the layer's output BEHIND its Dropout (drop_T output frames, batch-first
                    // element index as in the forward norm role): the same mask and scale, regenerated
                    const long long i0 = ((long long)b * drop_T + t / reduce) * H + c;
#pragma unroll
                    for (int e = 0; e < 8; ++e) dy[i][e] = ed_drop_keep(drop_seed, i0 + e, drop_thresh) ? dy[i][e] * drop_scale : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    dy[i][e] *= scale;
                    xh[i][e] = (z[e] - m) * rs;
                    const float g = dy[i][e] * gm[i][e];
                    s1 += g;
                    s2 += g * xh[i][e];
                    ag[i][e] += dy[i][e] * xh[i][e];
                    ab[i][e] += dy[i][e];
                }
            }
        }
        s1 = wave_sum(s1) / (float)H;
        s2 = wave_sum(s2) / (float)H;
        bf16_t* dr = dz + row * H;
#pragma unroll
        for (int i = 0; i < LNB_NB; ++i) {
            const int c = lane * 8 + 512 * i;
            if (c < H) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rs * (dy[i][e] * gm[i][e] - s1 - xh[i][e] * s2);
                ElemIO<bf16_t>::store_vec(dr + c, o);
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < LNB_NB; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[0][wave - 1][i * 8 + e][lane] = ag[i][e];
                red[1][wave - 1][i * 8 + e][lane] = ab[i][e];
            }
    }
    __syncthreads();
    if (wave == 0) {
        // this workgroup's partial sums: row blockIdx.x of part[.][2][H] (plain stores; the rows of
        // all launches of a layer are summed once at the end by stack_sum_parts_kernel)
        float* pg = part + (long long)blockIdx.x * 2 * H;
        float* pb = pg + H;
#pragma unroll
        for (int i = 0; i < LNB_NB; ++i) {
            const int c = lane * 8 + 512 * i;
            if (c < H) {
                float og[8], ob[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = i * 8 + e;
                    og[e] = ag[i][e] + red[0][0][k][lane] + red[0][1][k][lane] + red[0][2][k][lane];
                    ob[e] = ab[i][e] + red[1][0][k][lane] + red[1][1][k][lane] + red[1][2][k][lane];
                }
                *reinterpret_cast<float4*>(pg + c) = make_float4(og[0], og[1], og[2], og[3]);
                *reinterpret_cast<float4*>(pg + c + 4) = make_float4(og[4], og[5], og[6], og[7]);
                *reinterpret_cast<float4*>(pb + c) = make_float4(ob[0], ob[1], ob[2], ob[3]);
                *reinterpret_cast<float4*>(pb + c + 4) = make_float4(ob[4], ob[5], ob[6], ob[7]);
            }
        }
    }
}

// dgamma[c] += sum_r part[r][0][c];  dbeta[c] += sum_r part[r][1][c]
__global__ __launch_bounds__(256) void stack_sum_parts_kernel(const float* __restrict__ part, int rows,
                                                             int H, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta) {
    const int c = blockIdx.x * 256 + threadIdx.x;   // column of the [2][H] pair
    if (c >= 2 * H) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int r = blockIdx.y;
    const int stride = gridDim.y;
    for (; r + 3 * stride < rows; r += 4 * stride) {
        a0 += part[(long long)r * 2 * H + c];
        a1 += part[(long long)(r + stride) * 2 * H + c];
        a2 += part[(long long)(r + 2 * stride) * 2 * H + c];
        a3 += part[(long long)(r + 3 * stride) * 2 * H + c];
    }
    for (; r < rows; r += stride) a0 += part[(long long)r * 2 * H + c];
    atomicAdd(c < H ? dgamma + c : dbeta + (c - H), (a0 + a1) + (a2 + a3));
}

// =====================================================================================
// input LayerNorm (rnnt/models.py:124,132): x [B, T, D] batch-first (fp32 or bf16) ->
// out [T, B, D] time-major bf16.  One wave per row; D is small (240), scalar loads.
// =====================================================================================
template <typename T>
__global__ __launch_bounds__(256) void stack_input_norm_kernel(
    const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    bf16_t* __restrict__ out, float* __restrict__ mean_out, float* __restrict__ rstd_out, int B,
    int Tn, int D, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long rows = (long long)B * Tn;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
        const int b = (int)(row / Tn), t = (int)(row % Tn);
        const T* xr = x + row * D;
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += ElemIO<T>::load(xr + c);
        const float m = wave_sum(s) / (float)D;
        float v = 0.f;
        for (int c = lane; c < D; c += 64) {
            const float d = ElemIO<T>::load(xr + c) - m;
            v += d * d;
        }
        const float rs = rsqrtf(wave_sum(v) / (float)D + eps);
        if (lane == 0) {
            mean_out[row] = m;
            rstd_out[row] = rs;
        }
        bf16_t* o = out + ((long long)t * B + b) * D;
        for (int c = lane; c < D; c += 64)
            o[c] = f32_to_bf16((ElemIO<T>::load(xr + c) - m) * rs * gamma[c] + beta[c]);
    }
}

// dgamma[c] += sum_rows dX[t,b,c] * xhat[b,t,c];  dbeta[c] += sum_rows dX[t,b,c]
template <typename T>
__global__ __launch_bounds__(256) void stack_input_norm_bwd_kernel(
    const T* __restrict__ x, const bf16_t* __restrict__ dX, const float* __restrict__ mean_in,
    const float* __restrict__ rstd_in, float* __restrict__ dgamma, float* __restrict__ dbeta, int B,
    int Tn, int D, int rows_per_block) {
    const int c = threadIdx.x;
    if (c >= D) return;
    const long long rows = (long long)B * Tn;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(rows, r0 + rows_per_block);
    float ag = 0.f, ab = 0.f;
    for (long long row = r0; row < r1; ++row) {   // row enumerates time-major (t, b)
        const int t = (int)(row / B), b = (int)(row % B);
        const long long xrow = (long long)b * Tn + t;
        const float dy = bf16_to_f32(dX[row * D + c]);
        const float xh = (ElemIO<T>::load(x + xrow * D + c) - mean_in[xrow]) * rstd_in[xrow];
        ag += dy * xh;
        ab += dy;
    }
    atomicAdd(dgamma + c, ag);
    atomicAdd(dbeta + c, ab);
}

// Yx[0] <- bf16(h0), Cx[0] <- c0, fragment image of h0 (zeros when the states are null)
__global__ void stack_init_state_kernel(const float* __restrict__ h0, const float* __restrict__ c0,
                                        bf16_t* __restrict__ Yx0, float* __restrict__ Cx0,
                                        bf16_t* __restrict__ hfrag, int B, int H) {
    const int B16 = (B + 15) / 16 * 16;
    const long long n = (long long)B16 * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / H), j = (int)(i % H);
        const bool live = b < B;
        const bf16_t hv = f32_to_bf16((live && h0) ? h0[i] : 0.f);
        if (live) {
            Yx0[i] = hv;
            Cx0[i] = c0 ? c0[i] : 0.f;
        }
        hfrag[((((long long)(j >> 5) * (B16 >> 4) + (b >> 4)) * 64) + ((j & 31) >> 3) * 16 + (b & 15)) * 8 + (j & 7)] = hv;
    }
}

// the same for up to 8 layers in one launch (blockIdx.y = layer): six launches of ~5 us each sat in front of the forward
// pass's first recurrence launch
__global__ void stack_init_states_kernel(EdInitStates A) {
    const int l = blockIdx.y;
    const float* __restrict__ h0 = A.h0[l];
    const float* __restrict__ c0 = A.c0[l];
    bf16_t* __restrict__ Yx0 = A.Yx0[l];
    float* __restrict__ Cx0 = A.Cx0[l];
    bf16_t* __restrict__ hfrag = A.hfrag[l];
    const int B = A.B, H = A.H;
    const int B16 = (B + 15) / 16 * 16;
    const long long n = (long long)B16 * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / H), j = (int)(i % H);
        const bool live = b < B;
        const bf16_t hv = f32_to_bf16((live && h0) ? h0[i] : 0.f);
        if (live) {
            Yx0[i] = hv;
            Cx0[i] = c0 ? c0[i] : 0.f;
        }
        hfrag[((((long long)(j >> 5) * (B16 >> 4) + (b >> 4)) * 64) + ((j & 31) >> 3) * 16 + (b & 15)) * 8 + (j & 7)] = hv;
    }
}

}  // namespace

int ed_stack_launch_fwd(const EdFwdLaunch& L, hipStream_t s) {
    const int UB = L.H >> 4, RG = (L.B + 63) >> 6, RB = (L.B + 3) >> 2;
    const int grid = L.nstep * UB * RG + L.nnorm * RB;
    if (grid == 0) return ED_OK;
    hipLaunchKernelGGL(stack_fwd_kernel, dim3(grid), dim3(256), 0, s, L);
    ED_CHECK_LAUNCH("stack_fwd_kernel");
    return ED_OK;
}

int ed_stack_multi_norm(const EdChunkNorm* items, int n, int B, int H, float eps, hipStream_t s) {
    if (n <= 0) return ED_OK;
    ED_CHECK_ARG(n <= ED_STACK_MAX_SLOTS, "stack_multi_norm: too many items");
    MultiNormArgs M;
    int grid = 0;
    M.n = 0;
    for (int i = 0; i < n; ++i) {
        const EdChunkNorm& e = items[i];
        if (e.t1 <= e.t0) continue;
        ChunkNormArgs& a = M.a[M.n];
        a.Yx1 = e.Yx1; a.X = e.X; a.gamma = e.gamma; a.beta = e.beta; a.out = e.out; a.out_st = e.out_st;
        a.out_sb = e.out_sb; a.mean = e.mean; a.rstd = e.rstd; a.B = B; a.H = H; a.T = e.T; a.t0 = e.t0;
        a.reduce = e.reduce; a.eps = eps;
        a.drop_thresh = e.drop_thresh; a.drop_seed = e.drop_seed; a.drop_scale = e.drop_scale; a.drop_T = e.drop_T;
        M.first[M.n] = grid;
        const int frames_out = e.reduce == 1 ? e.t1 - e.t0 : (e.t1 - e.t0 + 1) / 2;
        grid += frames_out * ((B + 3) >> 2);
        ++M.n;
    }
    M.first[M.n] = grid;
    if (grid == 0) return ED_OK;
    hipLaunchKernelGGL(stack_multi_norm_kernel, dim3(grid), dim3(256), 0, s, M);
    ED_CHECK_LAUNCH("stack_multi_norm_kernel");
    return ED_OK;
}

int ed_stack_wait_counters(const unsigned* const* counters, const unsigned* targets, int n, unsigned* err, hipStream_t s) {
    if (n <= 0) return ED_OK;
    ED_CHECK_ARG(n <= WAIT_COUNTERS_MAX, "stack_wait_counters: too many counters");
    WaitCountersArgs a;
    for (int i = 0; i < n; ++i) {
        a.counter[i] = counters[i];
        a.target[i] = targets[i];
    }
    a.n = n;
    a.err = err;
    hipLaunchKernelGGL(stack_wait_counters_kernel, dim3(1), dim3(64), 0, s, a);
    ED_CHECK_LAUNCH("stack_wait_counters_kernel");
    return ED_OK;
}

int ed_stack_sk_supported(int B, int H) { return (B >= 1 && B <= 64 && H % 64 == 0 && H >= 64 && H <= 1024) ? 1 : 0; }

int ed_stack_pack_sk(const float* w_hh, bf16_t* out, int H, hipStream_t s) {
    hipLaunchKernelGGL(pack_whh_sk_kernel, dim3(ed_grid_for(4ll * H * H / 8, 256, 4096)), dim3(256), 0, s, w_hh, out, H);
    ED_CHECK_LAUNCH("pack_whh_sk_kernel");
    return ED_OK;
}

int ed_stack_launch_bwd_sk(const EdSkLaunch& L, hipStream_t s) {
    const int grid = L.nslot * (L.H >> 6) * 4;
    if (grid == 0) return ED_OK;
    if (L.trace) hipLaunchKernelGGL(stack_bwd_sk_kernel<true>, dim3(grid), dim3(256), SK_RING * SK_SLOT, s, L);
    else hipLaunchKernelGGL(stack_bwd_sk_kernel<false>, dim3(grid), dim3(256), SK_RING * SK_SLOT, s, L);
    ED_CHECK_LAUNCH("stack_bwd_sk_kernel");
    return ED_OK;
}

int ed_stack_lpw_supported(int B, int H) {
    const int UB = H >> 4, RG = (B + 63) >> 6;
    // at least four layer slots of one workgroup per CU must fit on the chip (256 CUs)
    return (H % 32 == 0 && (H >> 5) <= 4 * LPW_PER && UB * RG <= 64) ? 1 : 0;
}

int ed_stack_launch_fwd_lpw(const EdLpwLaunch& L, hipStream_t s) {
    const int UB = L.H >> 4, RG = (L.B + 63) >> 6;
    const int grid = L.nslot * UB * RG;
    if (grid == 0) return ED_OK;
#define ED_LPW_LAUNCH(TR, DPOLL) hipLaunchKernelGGL((stack_fwd_lpw_kernel<TR, DPOLL>), dim3(grid), dim3(256), 0, s, L)
    if (L.trace) { if (L.data_poll) ED_LPW_LAUNCH(true, true); else ED_LPW_LAUNCH(true, false); }
    else { if (L.data_poll) ED_LPW_LAUNCH(false, true); else ED_LPW_LAUNCH(false, false); }
#undef ED_LPW_LAUNCH
    ED_CHECK_LAUNCH("stack_fwd_lpw_kernel");
    return ED_OK;
}

int ed_stack_set_flag(unsigned* flag, hipStream_t s) {
    hipLaunchKernelGGL(stack_set_flag_kernel, dim3(1), dim3(64), 0, s, flag);
    ED_CHECK_LAUNCH("stack_set_flag_kernel");
    return ED_OK;
}

int ed_stack_launch_bwd(const EdBwdLaunch& L, hipStream_t s) {
    const int NB = L.H >> 5, RG = (L.B + 31) >> 5;
    const int grid = L.nstep * NB * RG;
    if (grid == 0) return ED_OK;
    hipLaunchKernelGGL(stack_bwd_kernel, dim3(grid), dim3(256), 0, s, L);
    ED_CHECK_LAUNCH("stack_bwd_kernel");
    return ED_OK;
}

int ed_stack_ln_bwd(const bf16_t* dout, long long dout_st, long long dout_sb, const bf16_t* y,
                    const bf16_t* res, const float* gamma, const float* mean, const float* rstd,
                    bf16_t* dz, float* part, int grid, int B, int H, int t0, int t1, int reduce,
                    hipStream_t s, unsigned drop_thresh, unsigned drop_seed, float drop_scale, int drop_T) {
    // fixed grid: every workgroup writes its partial row (zeros when it owns no rows)
    hipLaunchKernelGGL(stack_ln_bwd_kernel, dim3(grid), dim3(256), 0, s, dout, dout_st, dout_sb, y,
                       res, gamma, mean, rstd, dz, part, B, H, t0, max(t0, t1), reduce, drop_thresh, drop_seed,
                       drop_scale, drop_T);
    ED_CHECK_LAUNCH("stack_ln_bwd_kernel");
    return ED_OK;
}

int ed_stack_sum_parts(const float* part, int rows, int H, float* dgamma, float* dbeta, hipStream_t s) {
    if (rows <= 0) return ED_OK;
    const int gy = rows >= 64 ? 16 : 1;
    hipLaunchKernelGGL(stack_sum_parts_kernel, dim3((2 * H + 255) / 256, gy), dim3(256), 0, s, part,
                       rows, H, dgamma, dbeta);
    ED_CHECK_LAUNCH("stack_sum_parts_kernel");
    return ED_OK;
}

int ed_stack_input_norm(int x_dtype, const void* x, const float* gamma, const float* beta,
                        bf16_t* out, float* mean, float* rstd, int B, int T, int D, float eps,
                        hipStream_t s) {
    const int grid = ed_grid_for((long long)B * T, 4, 256 * 16);
    if (x_dtype == ED_F32)
        hipLaunchKernelGGL(stack_input_norm_kernel<float>, dim3(grid), dim3(256), 0, s,
                           (const float*)x, gamma, beta, out, mean, rstd, B, T, D, eps);
    else
        hipLaunchKernelGGL(stack_input_norm_kernel<bf16_t>, dim3(grid), dim3(256), 0, s,
                           (const bf16_t*)x, gamma, beta, out, mean, rstd, B, T, D, eps);
    ED_CHECK_LAUNCH("stack_input_norm_kernel");
    return ED_OK;
}

int ed_stack_input_norm_bwd(int x_dtype, const void* x, const bf16_t* dX, const float* mean,
                            const float* rstd, float* dgamma, float* dbeta, int B, int T, int D,
                            hipStream_t s) {
    const long long rows = (long long)B * T;
    const int rpb = 64;
    const int grid = (int)((rows + rpb - 1) / rpb);
    const int threads = (D + 63) / 64 * 64;
    if (x_dtype == ED_F32)
        hipLaunchKernelGGL(stack_input_norm_bwd_kernel<float>, dim3(grid), dim3(threads), 0, s,
                           (const float*)x, dX, mean, rstd, dgamma, dbeta, B, T, D, rpb);
    else
        hipLaunchKernelGGL(stack_input_norm_bwd_kernel<bf16_t>, dim3(grid), dim3(threads), 0, s,
                           (const bf16_t*)x, dX, mean, rstd, dgamma, dbeta, B, T, D, rpb);
    ED_CHECK_LAUNCH("stack_input_norm_bwd_kernel");
    return ED_OK;
}

int ed_stack_init_state(const float* h0, const float* c0, bf16_t* Yx0, float* Cx0, bf16_t* hfrag,
                        int B, int H, hipStream_t s) {
    const long long n = (long long)((B + 15) / 16 * 16) * H;
    hipLaunchKernelGGL(stack_init_state_kernel, dim3(ed_grid_for(n, 256)), dim3(256), 0, s, h0, c0,
                       Yx0, Cx0, hfrag, B, H);
    ED_CHECK_LAUNCH("stack_init_state_kernel");
    return ED_OK;
}

int ed_stack_init_states(const EdInitStates& A, hipStream_t s) {
    if (A.n <= 0) return ED_OK;
    ED_CHECK_ARG(A.n <= 8, "stack_init_states: too many layers in one call");
    const long long n = (long long)((A.B + 15) / 16 * 16) * A.H;
    hipLaunchKernelGGL(stack_init_states_kernel, dim3(ed_grid_for(n, 256), A.n), dim3(256), 0, s, A);
    ED_CHECK_LAUNCH("stack_init_states_kernel");
    return ED_OK;
}

int ed_stack_zero(void* p, size_t bytes, hipStream_t s) {
    ED_CHECK_HIP(hipMemsetAsync(p, 0, bytes, s));
    return ED_OK;
}
