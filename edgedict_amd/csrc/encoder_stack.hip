// Layer-pipelined LSTM encoder stack: stream scheduler + C ABI (bf16 throughput mode).
//
// Replaces the per-layer loop of ResLayerNormLSTM.forward (rnnt/models.py:55-75) plus the input
// LayerNorm of Encoder.forward (rnnt/models.py:124,132) and their autograd with ONE native call
// per direction.  The arithmetic per layer is unchanged (input product as a GEMM, recurrence,
// residual + LayerNorm + TimeReduction); what changes is the ORDER of the work:
//
//   * the recurrences of different layers are independent once their inputs exist, so layer l
//     runs `lag` launches behind layer l-1 and ONE launch of stack_fwd_kernel carries a time step
//     of every runnable layer (layers behind a 2x time reduction step every other launch) plus
//     the LayerNorms of the frames the previous launch finished: ~T0 + (L-1) lag launches
//     instead of sum_l T_l, each doing 4x the work of a per-layer step at the same latency;
//   * the input products X_l W_ih^T are GEMMs over CHUNKS of frames (time-major rows, so a chunk
//     is a contiguous operand) on a side stream, ordered against the recurrence stream with
//     events: chunk k of layer l+1 is multiplied while layer l works on chunk k+1;
//   * backward mirrors this top-down and in reverse time (BPTT launches; per chunk dX GEMM and
//     LayerNorm backward on the side stream); weight gradients run on a third, low-priority
//     stream as soon as a layer's BPTT is complete, under the BPTT of the layers below.
//
// The caller's stream is forked into the internal streams at entry and joined at exit, so the
// call is an ordinary stream-ordered operation for the caller (and for the caching allocator).
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "stack_kernels.hpp"

int ed_gemm_quiet_partials(int dtype_in, const void* A, long long lda, int a_kmajor, const void* B,
                           long long ldb, int b_kmajor, int M, int N, int K, int split_k,
                           int max_wg_per_cu, float* partials, int* slices, hipStream_t stream);   // gemm.hip

namespace {

// ---------------------------------------------------------------- weight images
// (Re-built after every optimiser step, in front of the next step's encoder: one 16-byte store per thread.  The
// element-per-thread versions took 10 us per W_hh image and 35 us for W_ih + its transpose - 0.39 ms per step for
// six layers, most of what separated the optimiser step from the first recurrence launch.)
__device__ __forceinline__ uint4 pack8(const float* v) {
    uint4 o;
    o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
    o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
    o.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
    o.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
    return o;
}
// W_ih row of interleaved gate column kap (stack_kernels.hpp ed_gate_col)
__device__ __forceinline__ int row_of_kap(int kap, int H) { return ((kap >> 4) & 3) * H + (kap >> 6) * 16 + (kap & 15); }

// wih_p[kappa(g,j)][k] = W_ih[g*H + j][k] (bf16);  bias_p[kappa] = b_ih + b_hh (fp32).  Any I (scalar).
__global__ void pack_wih_kernel(const float* __restrict__ W, const float* __restrict__ b_ih,
                                const float* __restrict__ b_hh, bf16_t* __restrict__ Wp,
                                bf16_t* __restrict__ Wt, float* __restrict__ bp, int H, int I) {
    const long long n = 4ll * H * I;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / I), k = (int)(i % I);
        const int g = row / H, j = row % H;
        const int kap = ed_gate_col(g, j);
        const bf16_t v = f32_to_bf16(W[i]);
        Wp[(long long)kap * I + k] = v;
        if (Wt) Wt[(long long)k * 4 * H + kap] = v;
        if (k == 0) bp[kap] = b_ih[row] + b_hh[row];
    }
}
// the same for I % 8 == 0: thread = (kap, 8 consecutive k)
__global__ void pack_wih8_kernel(const float* __restrict__ W, const float* __restrict__ b_ih,
                                 const float* __restrict__ b_hh, bf16_t* __restrict__ Wp, float* __restrict__ bp, int H,
                                 int I) {
    const int I8 = I >> 3;
    const long long n8 = 4ll * H * I8;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < n8; c += (long long)gridDim.x * blockDim.x) {
        const int kap = (int)(c / I8), k8 = (int)(c % I8);
        const int row = row_of_kap(kap, H);
        float v[8];
        const float4 a = *reinterpret_cast<const float4*>(W + (long long)row * I + k8 * 8);
        const float4 b = *reinterpret_cast<const float4*>(W + (long long)row * I + k8 * 8 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        *reinterpret_cast<uint4*>(Wp + (long long)kap * I + k8 * 8) = pack8(v);
        if (k8 == 0) bp[kap] = b_ih[row] + b_hh[row];
    }
}
// wih_t[k][kappa] = W_ih[row(kappa)][k]: 64 x 64 tiles through LDS (rows of W in, rows of the transpose out)
__global__ __launch_bounds__(256) void pack_wih_t_kernel(const float* __restrict__ W, bf16_t* __restrict__ Wt, int H,
                                                         int I) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64][64 + 8];                      // [k][kap], 144-byte rows: 16-byte aligned chunks
    const int kt = (I + 63) >> 6;
    const int kap0 = (blockIdx.x / kt) * 64, k0 = (blockIdx.x % kt) * 64;
    const int tid = threadIdx.x;
    for (int q = tid; q < 64 * 16; q += 256) {               // 64 rows x 16 float4
        const int r = q >> 4, c4 = (q & 15) * 4;
        const int row = row_of_kap(kap0 + r, H);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if ((I & 3) == 0 && k0 + c4 + 3 < I) {
            const float4 a = *reinterpret_cast<const float4*>(W + (long long)row * I + k0 + c4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        } else {
            for (int e = 0; e < 4; ++e)
                if (k0 + c4 + e < I) v[e] = W[(long long)row * I + k0 + c4 + e];
        }
        for (int e = 0; e < 4; ++e) tile[c4 + e][r] = f32_to_bf16(v[e]);
    }
    __syncthreads();
    for (int q = tid; q < 64 * 8; q += 256) {                // 64 k x 8 chunks of 8 kap
        const int k = q >> 3, c8 = (q & 7) * 8;
        if (k0 + k < I)
            *reinterpret_cast<uint4*>(Wt + (long long)(k0 + k) * 4 * H + kap0 + c8) = *reinterpret_cast<const uint4*>(&tile[k][c8]);
    }
}
// Fragment images of W_hh.  A wave's loads walk each image front to back (k-step major, the
// operands of one k-step adjacent), so a workgroup's slice is ONE sequential stream instead of
// several streams a power-of-two apart (which would camp on the same L2 channel).
// forward: frag[ub][ks][gate][lane][8]: n = lane & 15 -> W_hh row gate*H + ub*16 + n,
//          k = ks*32 + (lane >> 4)*8 + e
// thread = one 16-byte chunk (8 consecutive k of one row)
__global__ void pack_whh_fwd_kernel(const float* __restrict__ W, bf16_t* __restrict__ out, int H) {
    const long long n8 = 4ll * H * H / 8;
    const int KS = H >> 5;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < n8; c += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(c & 63);
        const long long blk = c >> 6;
        const int g = (int)(blk & 3), ks = (int)((blk >> 2) % KS), ub = (int)((blk >> 2) / KS);
        const int row = g * H + ub * 16 + (lane & 15);
        const int k = ks * 32 + (lane >> 4) * 8;
        float v[8];
        const float4 a = *reinterpret_cast<const float4*>(W + (long long)row * H + k);
        const float4 b = *reinterpret_cast<const float4*>(W + (long long)row * H + k + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        *reinterpret_cast<uint4*>(out + c * 8) = pack8(v);
    }
}
// backward: frag[nb32][ks][n2][lane][8]: n = lane & 15 -> hidden unit (nb32*2 + n2)*16 + n,
//           k = interleaved gate column ks*32 + (lane >> 4)*8 + e
// thread = one chunk: 8 consecutive gate columns = 8 consecutive W rows at one unit (16 lanes read 64 contiguous bytes
// of each)
__global__ void pack_whh_bwd_kernel(const float* __restrict__ W, bf16_t* __restrict__ out, int H) {
    const long long n8 = 4ll * H * H / 8;
    const int KS = H >> 3;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < n8; c += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(c & 63);
        const long long blk = c >> 6;
        const int n2 = (int)(blk & 1), ks = (int)((blk >> 1) % KS), nb = (int)((blk >> 1) / KS);
        const int unit = (nb * 2 + n2) * 16 + (lane & 15);
        const int kap = ks * 32 + (lane >> 4) * 8;
        const float* src = W + (long long)row_of_kap(kap, H) * H + unit;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[(long long)e * H];
        *reinterpret_cast<uint4*>(out + c * 8) = pack8(v);
    }
}
// dst[g*H + j][k] (+)= sum_s src[s][kappa(g,j)][k]   (slices of a quiet split-K product; dst2 gets
// the same values: the two LSTM bias gradients are equal)
__global__ void unpermute_rows_kernel(const float* __restrict__ src, long long slice_stride, int S,
                                      float* __restrict__ dst, float* __restrict__ dst2, int H, int K,
                                      int accumulate) {
    const long long n = 4ll * H * K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / K), k = (int)(i % K);
        const long long o = (long long)ed_gate_col(row / H, row % H) * K + k;
        float v = 0.f;
        for (int sl = 0; sl < S; ++sl) v += src[sl * slice_stride + o];
        dst[i] = accumulate ? dst[i] + v : v;
        if (dst2) dst2[i] = accumulate ? dst2[i] + v : v;
    }
}

// the same, four columns per thread (K % 4 == 0, 16-byte aligned buffers): same sums in the same order
__global__ void unpermute_rows4_kernel(const float* __restrict__ src, long long slice_stride, int S,
                                       float* __restrict__ dst, float* __restrict__ dst2, int H, int K,
                                       int accumulate) {
    const int K4 = K >> 2;
    const long long n4 = 4ll * H * K4;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < n4; c += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(c / K4), k = (int)(c % K4) * 4;
        const long long o = (long long)ed_gate_col(row / H, row % H) * K + k;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int sl = 0; sl < S; ++sl) {
            const float4 p = *reinterpret_cast<const float4*>(src + sl * slice_stride + o);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        const long long i = (long long)row * K + k;
        if (accumulate) {
            const float4 d = *reinterpret_cast<const float4*>(dst + i);
            *reinterpret_cast<float4*>(dst + i) = make_float4(d.x + v.x, d.y + v.y, d.z + v.z, d.w + v.w);
            if (dst2) {
                const float4 e = *reinterpret_cast<const float4*>(dst2 + i);
                *reinterpret_cast<float4*>(dst2 + i) = make_float4(e.x + v.x, e.y + v.y, e.z + v.z, e.w + v.w);
            }
        } else {
            *reinterpret_cast<float4*>(dst + i) = v;
            if (dst2) *reinterpret_cast<float4*>(dst2 + i) = v;
        }
    }
}
int unpermute_rows(const float* src, long long slice_stride, int S, float* dst, float* dst2, int H, int K, int accumulate,
                   hipStream_t s) {
    const bool vec = K % 4 == 0 && slice_stride % 4 == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0 &&
                     (uintptr_t)dst2 % 16 == 0;
    if (vec)
        hipLaunchKernelGGL(unpermute_rows4_kernel, dim3(ed_grid_for(4ll * H * K / 4, 256, 4096)), dim3(256), 0, s, src,
                           slice_stride, S, dst, dst2, H, K, accumulate);
    else
        hipLaunchKernelGGL(unpermute_rows_kernel, dim3(ed_grid_for(4ll * H * K, 256, 4096)), dim3(256), 0, s, src,
                           slice_stride, S, dst, dst2, H, K, accumulate);
    ED_CHECK_LAUNCH("unpermute_rows_kernel");
    return ED_OK;
}

// ---------------------------------------------------------------- per-device runtime
struct Runtime {
    // ONE call at a time per device (the event pool, the timing record and the stamp buffers belong to a call);
    // calls on different devices - one host thread per GPU, nn.DataParallel style - do not serialise
    std::mutex mu;
    hipStream_t R = nullptr, W = nullptr;
    int prio_hi = 0;
    // timing of the recurrence stream's launch sequence of the last forward / backward call
    hipEvent_t tev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    int tlaunches[2] = {0, 0};
    int tkind[2] = {0, 0}, tsteps[2] = {0, 0};   // recurrence kernel of the last pass (edgedict_stack_last_mode)
    // opt-in (edgedict_stack_time_launches): every wavefront launch of a call stamps its first workgroup's
    // start and its last workgroup's end (100 MHz clock) into its own slot of a device buffer - the kernels'
    // own durations, without the gaps between launches that the span above includes
    bool time_each = false;
    static constexpr int STAMP_SLOTS = 4096;
    unsigned long long* stamps[2] = {nullptr, nullptr};   // [STAMP_SLOTS][2] per direction
    int stamp_used[2] = {0, 0};
    unsigned long long* stamp_slot(int i, hipStream_t s) {
        if (!time_each) return nullptr;
        if (!stamps[i] && hipMalloc((void**)&stamps[i], (size_t)STAMP_SLOTS * 16) != hipSuccess) {
            stamps[i] = nullptr;
            return nullptr;
        }
        if (stamp_used[i] == 0) {   // start-of-call: min slots to all-ones, max slots to zero
            if (hipMemsetAsync(stamps[i], 0, (size_t)STAMP_SLOTS * 16, s) != hipSuccess) return nullptr;
            if (hipMemset2DAsync(stamps[i], 16, 0xff, 8, STAMP_SLOTS, s) != hipSuccess) return nullptr;
        }
        if (stamp_used[i] >= STAMP_SLOTS) return nullptr;
        return stamps[i] + 2 * (size_t)stamp_used[i]++;
    }
    // give-up code of the weights-stationary kernels: pinned host word the device writes on failure
    unsigned* wsr_err_host = nullptr;
    unsigned* wsr_err_dev = nullptr;
    // experiment streams, created on first use (and therefore AFTER the three above)
    hipStream_t lazy(hipStream_t& s) {
        if (!s && hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio_hi) != hipSuccess) s = nullptr;
        return s;
    }
    hipStream_t S[ED_STACK_MAX_SLOTS] = {};   // one side stream per layer (chunk GEMMs / LayerNorm backward)
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    hipEvent_t get() {
        if (used == pool.size()) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
            pool.push_back(e);
        }
        return pool[used++];
    }
};
std::mutex g_rt_mu;           // the registry below only, never held across a call
std::vector<Runtime*> g_rt;   // indexed by device ordinal

Runtime* runtime_for_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> registry(g_rt_mu);
    if ((int)g_rt.size() <= dev) g_rt.resize(dev + 1, nullptr);
    if (!g_rt[dev]) {
        Runtime* r = new Runtime();
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = hi = 0;   // lo = least, hi = greatest priority
        const char* e = getenv("EDGEDICT_AUX_PRIORITY");
        const bool aux_high = e && e[0] == '1';
        // HIP maps streams onto 4 hardware queues round-robin in creation order: create ONLY the
        // three streams the default schedule uses (recurrence, auxiliary, chunk GEMMs; with the
        // caller's stream that is one queue each).  Streams sharing a queue serialise: creating
        // one more stream before these was measured to cost 13 ms per training step, and a process
        // that creates OTHER streams first (torch's stream pool, RCCL at its first collective) pays
        // 32-65 ms instead of 27 per step (tools/stream_order_probe.py; the priority class of S makes
        // no difference) - so hosts call edgedict_aux_stream() right after selecting the device,
        // before anything else creates streams (trainer.TrainEngine, bench.py do).
        r->prio_hi = hi;
        bool ok = hipStreamCreateWithPriority(&r->R, hipStreamNonBlocking, hi) == hipSuccess &&
                  hipStreamCreateWithPriority(&r->W, hipStreamNonBlocking, aux_high ? hi : lo) == hipSuccess &&
                  hipStreamCreateWithPriority(&r->S[0], hipStreamNonBlocking, hi) == hipSuccess;
        if (!ok) {
            delete r;
            return nullptr;
        }
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                if (hipEventCreate(&r->tev[i][j]) != hipSuccess) r->tev[i][j] = nullptr;
        if (hipHostMalloc((void**)&r->wsr_err_host, 64, hipHostMallocMapped) == hipSuccess) {
            r->wsr_err_host[0] = r->wsr_err_host[1] = r->wsr_err_host[2] = 0;
            if (hipHostGetDevicePointer((void**)&r->wsr_err_dev, r->wsr_err_host, 0) != hipSuccess)
                r->wsr_err_dev = nullptr;
        }
        g_rt[dev] = r;
    }
    return g_rt[dev];
}

struct Geom {
    int T, f, m, cf, off, nchunks;
};

size_t align256(size_t x) { return (x + 255) / 256 * 256; }
constexpr int LNB_GRID = 128;       // workgroups (= partial rows) per chunk call of the LayerNorm backward
constexpr int LNB_GRID_TOP = 512;   // ... of the top layer's all-frames call

struct WsLayout {
    std::vector<size_t> frag0, frag1, dC, lnpart;   // per layer
    std::vector<size_t> himg;                       // per layer: one h fragment image per frame (launch-persistent forward)
    size_t himg_stride = 0;
    std::vector<size_t> gimg;                       // per layer: one dG fragment image per frame (launch-persistent BPTT)
    size_t gimg_stride = 0;
    std::vector<size_t> skpart;                     // per layer: partial sums of the split-K BPTT [2][H/64][4][64][64] f32
    size_t tmpW = 0, tmpB = 0, dX0 = 0, wsr_sync = 0, total = 0;
};
// sync region of the workspace, in 4-byte words: chunk flags of the forward / backward pass [8 layers][512 chunks] each,
// arrival counters of the launch-persistent forward and of the split-K BPTT's layer-wide fallback [8 layers] lines of 256
// bytes each, from 64 KB the unit-block and quarter counters of the split-K BPTT [8 layers][SK_CNT_LINES] lines
constexpr size_t SYNC_FFLAG = 0, SYNC_BFLAG = 8 * 512, SYNC_FCNT = 2 * 8 * 512;
constexpr size_t SYNC_BCNT = SYNC_FCNT + (size_t)8 * LPW_CNT_STRIDE;
constexpr size_t SYNC_GCNT = 16 * 1024;
constexpr size_t WSR_SYNC_BYTES = (SYNC_GCNT + (size_t)8 * SK_CNT_LINES * 64) * 4;
static_assert(SYNC_BCNT + (size_t)8 * LPW_CNT_STRIDE <= SYNC_GCNT, "sync region layout");

WsLayout ws_layout(const edgedict_stack_desc_t* d) {
    WsLayout w;
    const size_t B16 = (size_t)(d->B + 15) / 16 * 16;
    size_t off = 0;
    size_t maxK = d->H;
    for (int l = 0; l < d->L; ++l) {
        // ping-pong fragment images: h (forward, B16 x H) or dG (backward, B16 x 4H)
        const size_t fb = align256(B16 * 4 * d->H * sizeof(bf16_t));
        w.frag0.push_back(off); off += fb;
        w.frag1.push_back(off); off += fb;
        if ((size_t)d->layers[l].I > maxK) maxK = d->layers[l].I;
    }
    // running dL/dc of every layer, contiguous: the backward pass zeroes them with ONE memset (nine memsets of 2-3 us,
    // ~6 us apart, sat in front of the BPTT's first launch)
    for (int l = 0; l < d->L; ++l) {
        w.dC.push_back(off);
        off += align256((size_t)d->B * d->H * sizeof(float));
    }
    // LayerNorm-backward partial sums: LNB_GRID rows per chunk call + LNB_GRID_TOP for the top layer
    {
        int f = 1;
        std::vector<int> fl(d->L);
        for (int l = d->L - 1; l >= 0; --l) { f *= max(1, d->layers[l].reduce); fl[l] = f; }
        for (int l = 0; l < d->L; ++l) {
            const int cf = max(1, d->chunk) * fl[l];
            const int nch = (d->layers[l].T + cf - 1) / cf;
            w.lnpart.push_back(off);
            off += align256((size_t)(nch * LNB_GRID + LNB_GRID_TOP) * 2 * d->H * sizeof(float));
        }
    }
    w.tmpW = off; off += align256((size_t)8 * 4 * d->H * maxK * sizeof(float));   // up to 8 K slices
    w.tmpB = off; off += align256((size_t)4 * d->H * sizeof(float));
    w.dX0 = off; off += align256((size_t)d->T0 * d->B * d->I0 * sizeof(bf16_t));
    w.wsr_sync = off; off += WSR_SYNC_BYTES;
    // one h image per frame and layer (stack_kernels.hpp EdLpwSlot::img): (T + 1) x B16 x H bf16 - 51 MB for an
    // E6D2 layer at 15 s; 288 GB of HBM is what makes "never write an address twice" affordable
    w.himg_stride = align256(B16 * d->H * sizeof(bf16_t));
    for (int l = 0; l < d->L; ++l) {
        w.himg.push_back(off);
        off += (size_t)(d->layers[l].T + 1) * w.himg_stride;
    }
    // ... and one dG image per frame for the launch-persistent BPTT: T x B16 x 4H bf16 (205 MB per full-rate E6D2 layer)
    // (nothing of it for a descriptor that promises no backward pass: 0.8 GB at E6D2 / 15 s / B = 64)
    const bool infer = (d->flags & EDGEDICT_STACK_INFERENCE) != 0;
    w.gimg_stride = align256(B16 * 4 * d->H * sizeof(bf16_t));
    for (int l = 0; l < d->L; ++l) {
        w.gimg.push_back(off);
        if (!infer) off += (size_t)(d->layers[l].T + 1) * w.gimg_stride;
    }
    for (int l = 0; l < d->L; ++l) {
        w.skpart.push_back(off);
        if (!infer) off += align256((size_t)2 * (d->H / 64 + 1) * 4 * 64 * 64 * sizeof(float));
    }
    w.total = off;
    return w;
}

int validate(const edgedict_stack_desc_t* d, std::vector<Geom>& g, bool backward) {
    ED_CHECK_ARG(d && d->layers, "encoder_stack: null descriptor");
    ED_CHECK_ARG(d->L >= 1 && d->L <= ED_STACK_MAX_SLOTS, "encoder_stack: 1..%d layers supported (got %d)", ED_STACK_MAX_SLOTS, d->L);
    ED_CHECK_ARG(d->B >= 1 && d->H >= 32 && d->H % 32 == 0 && d->H <= 2048, "encoder_stack: need B >= 1, H %% 32 == 0, H <= 2048 (B=%d H=%d)", d->B, d->H);
    ED_CHECK_ARG(d->chunk >= 1, "encoder_stack: chunk must be >= 1");
    ED_CHECK_ARG(d->T0 >= 1 && d->I0 >= 8 && d->I0 % 8 == 0 && d->I0 <= 1024, "encoder_stack: bad input geometry (T0=%d I0=%d)", d->T0, d->I0);
    ED_CHECK_ARG(d->x && d->in_gamma && d->in_beta && d->in_mean && d->in_rstd && d->out && d->ws, "encoder_stack: null pointer in descriptor");
    g.resize(d->L);
    int f = 1;
    for (int l = d->L - 1; l >= 0; --l) {
        const edgedict_stack_layer_t& y = d->layers[l];
        ED_CHECK_ARG(y.reduce == 1 || y.reduce == 2, "encoder_stack: layer %d: time reduction must be 1 or 2", l);
        f *= y.reduce;
        g[l].f = f;
        g[l].T = y.T;
    }
    int T = d->T0, I = d->I0;
    for (int l = 0; l < d->L; ++l) {
        const edgedict_stack_layer_t& y = d->layers[l];
        ED_CHECK_ARG(y.T == T && y.I == I, "encoder_stack: layer %d geometry (T=%d I=%d) does not follow from the layers above (T=%d I=%d)", l, y.T, y.I, T, I);
        ED_CHECK_ARG(!y.residual || y.I == d->H, "encoder_stack: layer %d: residual needs I == H", l);
        ED_CHECK_ARG(y.drop_p >= 0.f && y.drop_p < 1.f, "encoder_stack: layer %d: dropout probability must be in [0, 1)", l);
        ED_CHECK_ARG(y.wih_p && y.bias_p && y.whh_f && y.ln_gamma && y.ln_beta && y.X && y.G && y.Yx && y.Cx && y.mean && y.rstd, "encoder_stack: layer %d: null pointer", l);
        if (backward)
            ED_CHECK_ARG(y.whh_b && y.dZ && (l == 0 || y.dX) && y.dW_ih && y.dW_hh && y.db && y.dgamma && y.dbeta, "encoder_stack: layer %d: null backward pointer", l);
        g[l].m = g[0].f / g[l].f;
        g[l].cf = d->chunk * g[l].f;
        g[l].nchunks = (y.T + g[l].cf - 1) / g[l].cf;
        T = (T + y.reduce - 1) / y.reduce;
        I = d->H;
    }
    ED_CHECK_ARG(d->ws_bytes >= ws_layout(d).total, "encoder_stack: workspace too small");
    return ED_OK;
}

// ---- dynamic wavefront schedule ----------------------------------------------------------
// A layer steps in launch w when (a) the chunk its next step opens has been enqueued on the side
// stream at least `margin` launches ago (the product then has had time to run: the recurrence
// stream's event wait does not stall), and (b) its pacing allows it.  Pacing: a layer behind a time
// reduction needs one step per m_l launches only WHILE faster layers run beside it (every launch
// then carries <= one step per full-rate layer plus every m-th slower one, alternating by layer
// parity: the chip holds 4 layer-steps at one workgroup per CU); when no faster layer is running -
// the tail of the forward pass, the head of the backward pass - it steps in every launch, which
// removes most of the pipeline skew (E6D2: 547 -> ~490 forward launches, 546 -> ~490 backward).
struct Pace {
    // period of layer l given the smallest m among the layers j <= l that are "running"
    static bool allows(int w, int l, int m_l, int m_min_running) {
        const int period = max(1, m_l / max(1, m_min_running));
        return period <= 1 || ((w + period * 1024 - (l % period)) % period) == 0;
    }
};

int margin_launches(const edgedict_stack_desc_t* d, const std::vector<Geom>& g, bool backward) {
    // d->lag keeps its meaning "launches between a producer's and a consumer's first step of a chunk"
    // (P = launches per chunk): forward default P + 3, backward P + 5 -> margins 3 and 6 launches
    // after the chunk's side-stream work was enqueued.  Measured (E6D2): forward 8.15 ms at P+3 vs
    // 8.28 at P+5; backward 13.7 at P+5 vs 14.3 at P+3 (its chunk chain is longer: dX product +
    // LayerNorm backward); margins of 2 / 3 lose 1 ms.
    const int P = d->chunk * g[0].f;
    if (d->lag > 0) return max(1, d->lag - P + (backward ? 1 : 0));
    static const int mf = [] { const char* e = getenv("EDGEDICT_STACK_MARGIN_F"); return e ? atoi(e) : 0; }();
    static const int mb = [] { const char* e = getenv("EDGEDICT_STACK_MARGIN_B"); return e ? atoi(e) : 0; }();
    if (backward ? mb > 0 : mf > 0) return backward ? mb : mf;
    return backward ? 6 : 3;
}

#define ED_TRY(expr)            \
    do {                        \
        int st__ = (expr);      \
        if (st__ != ED_OK) return st__; \
    } while (0)

// ---- dry run: the schedulers below also run WITHOUT a device (edgedict_stack_schedule): every
// device call is skipped and the launch index of each layer-step / chunk product is recorded, so the
// host logic (dependencies, pacing, slot counts) is testable on a CPU-only machine.
struct ScheduleTrace {
    int32_t* step_launch = nullptr;      // [sum_l T_l], layer-major: launch that carries step (l, t)
    int32_t* chunk_enqueued = nullptr;   // [sum_l nchunks_l]: launches issued when the chunk's side-stream
                                         // product was enqueued (-1: available from the start)
    std::vector<int> toff, coff;
    int launches = 0, max_slots = 0;
};
thread_local ScheduleTrace* g_trace = nullptr;
#define ED_DEV(expr)                      \
    do {                                  \
        if (!g_trace) ED_TRY(expr);       \
    } while (0)

struct Streams {
    hipStream_t C, R, R2, W;
    int split;   // layers >= split run their recurrence on R2 (L = never)
    int which(int l) const { return l >= split ? 1 : 0; }
    hipStream_t RS(int l) const { return which(l) ? R2 : R; }
    hipStream_t S[ED_STACK_MAX_SLOTS];
    bool serial;
    Runtime* rt;
    std::unique_lock<std::mutex> lock;     // rt->mu for the duration of the call (open_streams)
    // order `waiter` after everything enqueued so far on `src`
    // share(src): ONE record on `src` that every chain(src, ...) reuses until unshare() - the side work of all the
    // chunks a BPTT launch completes hangs on the same point of the recurrence stream, and each record on that
    // stream costs the next launch ~3.5 us
    hipEvent_t shared_ev = nullptr;
    hipStream_t shared_src = nullptr;
    int share(hipStream_t src) {
        if (serial) return ED_OK;
        shared_ev = rt->get();
        ED_CHECK_ARG(shared_ev != nullptr, "encoder_stack: event creation failed");
        ED_CHECK_HIP(hipEventRecord(shared_ev, src));
        shared_src = src;
        return ED_OK;
    }
    void unshare() { shared_ev = nullptr; shared_src = nullptr; }
    int chain(hipStream_t src, hipStream_t waiter) {
        if (serial || src == waiter) return ED_OK;
        if (shared_ev && src == shared_src) {
            ED_CHECK_HIP(hipStreamWaitEvent(waiter, shared_ev, 0));
            return ED_OK;
        }
        hipEvent_t e = rt->get();
        ED_CHECK_ARG(e != nullptr, "encoder_stack: event creation failed");
        ED_CHECK_HIP(hipEventRecord(e, src));
        ED_CHECK_HIP(hipStreamWaitEvent(waiter, e, 0));
        return ED_OK;
    }
    int record(hipEvent_t& e, hipStream_t src) {
        if (serial) return ED_OK;
        e = rt->get();
        ED_CHECK_ARG(e != nullptr, "encoder_stack: event creation failed");
        ED_CHECK_HIP(hipEventRecord(e, src));
        return ED_OK;
    }
    int wait(hipStream_t waiter, hipEvent_t e) {
        if (serial || !e) return ED_OK;
        ED_CHECK_HIP(hipStreamWaitEvent(waiter, e, 0));
        return ED_OK;
    }
};

int open_streams(const edgedict_stack_desc_t* d, void* stream_, Streams& st) {
    st.C = (hipStream_t)stream_;
    st.serial = (d->flags & EDGEDICT_STACK_SERIAL) != 0 || g_trace != nullptr;
    st.rt = nullptr;
    if (st.serial) {
        st.R = st.R2 = st.W = st.C;
        for (auto& x : st.S) x = st.C;
        return ED_OK;
    }
    st.rt = runtime_for_current_device();
    ED_CHECK_ARG(st.rt != nullptr, "encoder_stack: could not create the internal streams");
    st.lock = std::unique_lock<std::mutex>(st.rt->mu);
    st.rt->used = 0;   // recycle the event pool (waits capture an event's state when enqueued)
    st.R = st.rt->R;
    st.R2 = st.rt->R;
    for (int i = 0; i < ED_STACK_MAX_SLOTS; ++i) {
        st.S[i] = st.rt->S[0];
        if ((d->flags & EDGEDICT_STACK_SIDE_STREAM_PER_LAYER) && i > 0) {
            st.S[i] = st.rt->lazy(st.rt->S[i]);
            ED_CHECK_ARG(st.S[i] != nullptr, "encoder_stack: stream creation failed");
        }
    }
    st.W = st.rt->W;
    return ED_OK;
}

// dropout behind layer y's LayerNorm (+ TimeReduction): mask threshold (0 = off), scale, frames of the layer's output
struct DropCfg { unsigned thresh, seed; float scale; int T_out; };
inline DropCfg drop_of(const edgedict_stack_layer_t& y) {
    DropCfg c;
    c.thresh = y.drop_p > 0.f ? ed_drop_thresh(y.drop_p) : 0u;
    c.seed = y.drop_seed;
    c.scale = y.drop_p > 0.f ? 1.f / (1.f - y.drop_p) : 1.f;
    c.T_out = (y.T + y.reduce - 1) / y.reduce;
    return c;
}

inline bf16_t* bptr(void* p) { return reinterpret_cast<bf16_t*>(p); }
inline const bf16_t* bptr(const void* p) { return reinterpret_cast<const bf16_t*>(p); }

// G_l[chunk k] = X_l[chunk k] W_ih^T + b on stream s
int input_gemm(const edgedict_stack_desc_t* d, const std::vector<Geom>& g, int l, int k, hipStream_t s) {
    const edgedict_stack_layer_t& y = d->layers[l];
    const int t0 = k * g[l].cf, t1 = min(y.T, t0 + g[l].cf);
    const long long r0 = (long long)t0 * d->B;
    return edgedict_gemm(ED_BF16, ED_BF16, bptr(y.X) + r0 * y.I, y.I, 1, y.wih_p, y.I, 1,
                         bptr(y.G) + r0 * 4 * d->H, 4ll * d->H, (t1 - t0) * d->B, 4 * d->H, y.I,
                         y.bias_p, nullptr, 0, 1, s);
}

}  // namespace

extern "C" void* edgedict_aux_stream(int which) {
    Runtime* r = runtime_for_current_device();
    if (!r) {
        ed_set_error("aux_stream: could not create the internal streams");
        return nullptr;
    }
    return which == 0 ? (void*)r->R : which == 1 ? (void*)r->S[0] : (void*)r->W;
}

// Which of the library's internal streams of the current device still have work enqueued (hipStreamQuery): bit 0 the
// recurrence stream, bit 1 the chunk-GEMM stream, bit 2 the auxiliary (weight-gradient) stream, bits 3.. the lazily
// created per-layer side streams.  Every entry point joins the streams it used into the caller's stream before it
// returns and every host-side borrower of the auxiliary stream joins it too (side.py), so once the CALLER's stream is
// idle a set bit is work that nothing orders behind - the caching allocator may hand its buffers to somebody else.
// Never creates the streams (0 if the runtime of this device does not exist yet).
extern "C" int edgedict_streams_busy(unsigned* mask) {
    ED_CHECK_ARG(mask != nullptr, "streams_busy: null mask");
    *mask = 0;
    int dev = 0;
    ED_CHECK_HIP(hipGetDevice(&dev));
    Runtime* r = nullptr;
    {
        std::lock_guard<std::mutex> registry(g_rt_mu);
        if ((int)g_rt.size() > dev) r = g_rt[dev];
    }
    if (!r) return ED_OK;
    auto busy = [&](hipStream_t s, int bit) -> int {
        if (!s) return ED_OK;
        const hipError_t e = hipStreamQuery(s);
        if (e == hipErrorNotReady) { *mask |= 1u << bit; (void)hipGetLastError(); return ED_OK; }
        ED_CHECK_HIP(e);
        return ED_OK;
    };
    ED_TRY(busy(r->R, 0));
    ED_TRY(busy(r->S[0], 1));
    ED_TRY(busy(r->W, 2));
    for (int i = 1; i < ED_STACK_MAX_SLOTS; ++i) ED_TRY(busy(r->S[i], 2 + i));
    return ED_OK;
}

extern "C" size_t edgedict_stack_struct_bytes(int which) {
    return which == 0 ? sizeof(edgedict_stack_layer_t) : sizeof(edgedict_stack_desc_t);
}

extern "C" size_t edgedict_stack_workspace_bytes(const edgedict_stack_desc_t* d) {
    if (!d || !d->layers || d->L < 1) return 0;
    return ws_layout(d).total;
}

extern "C" int edgedict_stack_pack_weights(const float* w_ih, const float* w_hh, const float* b_ih,
                                           const float* b_hh, int H, int I, void* wih_p, void* wih_t,
                                           float* bias_p, void* whh_f, void* whh_b, void* stream_) {
    ED_CHECK_ARG(H >= 32 && H % 32 == 0 && I >= 1, "stack_pack_weights: need H %% 32 == 0 (H=%d I=%d)", H, I);
    ED_CHECK_ARG(w_ih && w_hh, "stack_pack_weights: null weight pointer");
    ED_CHECK_ARG(wih_p || wih_t || whh_f || whh_b, "stack_pack_weights: no output image requested");
    ED_CHECK_ARG(!wih_p || (b_ih && b_hh && bias_p), "stack_pack_weights: wih_p comes with the packed bias (b_ih, b_hh, bias_p)");
    ED_CHECK_ARG((uintptr_t)w_hh % 16 == 0 && (uintptr_t)whh_f % 16 == 0 && (uintptr_t)whh_b % 16 == 0 &&
                     (uintptr_t)wih_t % 16 == 0,
                 "stack_pack_weights: W_hh and the images must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream_;
    // every output is optional: the images of the forward pass (wih_p + bias_p, whh_f) can be rebuilt in front of it and
    // those only the backward pass reads (wih_t, whh_b; edgedict_stack_pack_sk) behind it, off the critical path
    if (wih_p) {
        if (I % 8 == 0 && (uintptr_t)w_ih % 16 == 0 && (uintptr_t)wih_p % 16 == 0) {
            hipLaunchKernelGGL(pack_wih8_kernel, dim3(ed_grid_for(4ll * H * I / 8, 256, 4096)), dim3(256), 0, s,
                               w_ih, b_ih, b_hh, (bf16_t*)wih_p, bias_p, H, I);
            ED_CHECK_LAUNCH("pack_wih8_kernel");
        } else {
            hipLaunchKernelGGL(pack_wih_kernel, dim3(ed_grid_for(4ll * H * I, 256, 4096)), dim3(256), 0, s,
                               w_ih, b_ih, b_hh, (bf16_t*)wih_p, (bf16_t*)nullptr, bias_p, H, I);
            ED_CHECK_LAUNCH("pack_wih_kernel");
        }
    }
    if (wih_t) {
        hipLaunchKernelGGL(pack_wih_t_kernel, dim3((4 * H / 64) * ((I + 63) / 64)), dim3(256), 0, s, w_ih,
                           (bf16_t*)wih_t, H, I);
        ED_CHECK_LAUNCH("pack_wih_t_kernel");
    }
    if (whh_f) {
        hipLaunchKernelGGL(pack_whh_fwd_kernel, dim3(ed_grid_for(4ll * H * H / 8, 256, 4096)), dim3(256), 0,
                           s, w_hh, (bf16_t*)whh_f, H);
        ED_CHECK_LAUNCH("pack_whh_fwd_kernel");
    }
    if (whh_b) {
        hipLaunchKernelGGL(pack_whh_bwd_kernel, dim3(ed_grid_for(4ll * H * H / 8, 256, 4096)), dim3(256),
                           0, s, w_hh, (bf16_t*)whh_b, H);
        ED_CHECK_LAUNCH("pack_whh_bwd_kernel");
    }
    return ED_OK;
}

namespace {

long long* g_wsr_trace = nullptr;   // debug: device buffer registered by edgedict_stack_wsr_set_trace (tools/lpw_trace.py, tools/sk_trace.py)

int device_cus() {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
    return n;
}
bool lpw_data_poll() {
    // default: no counter on the dependency chain - the images are filled before the pass and the readers validate what
    // they gather; 0: the readers poll the arrival counters (round 3's protocol: 7.9 instead of 7.1 us per step)
    const char* e = getenv("EDGEDICT_LPW_POLL");
    return e ? atoi(e) != 0 : true;
}
bool poison_on() {
    const char* e = getenv("EDGEDICT_STACK_POISON");      // read per call: tests switch it inside one process
    return e && atoi(e) != 0;
}
int ed_stack_fill(void* p, int byte, size_t bytes, hipStream_t s) {
    ED_CHECK_HIP(hipMemsetAsync(p, byte, bytes, s));
    return ED_OK;
}


// Forward pass with the launch-persistent step kernel (stack_kernels.hip, stack_fwd_lpw_kernel): the
// wavefront schedule of edgedict_stack_forward in MACRO-steps of `nsub` consecutive time steps.  Launch w
// carries, for every runnable layer, the next `nsub` steps of that layer (never across a chunk boundary); the
// layers' workgroups keep W_hh in registers for the launch and meet step by step through arrival counters.
// After the launch the side stream normalises the frames each layer finished (ONE launch for all layers) and,
// for a layer whose chunk is now complete, multiplies it into the next layer's gates and sets that chunk's
// flag.  Runnable = the chunk the macro-step opens was enqueued at least `margin` launches ago; layers behind
// a time reduction are paced by launch parity while a faster layer runs (Pace), and a launch holds at most
// one workgroup per CU (max_slots).  Called with the prologue done and the internal streams forked.
int forward_lpw(const edgedict_stack_desc_t* d, const std::vector<Geom>& g, Streams& st, const WsLayout& wl,
                int nsub, bool soft) {
    const int B = d->B, H = d->H, L = d->L;
    const long long BH = (long long)B * H;
    char* ws = (char*)d->ws;
    unsigned* fflag = reinterpret_cast<unsigned*>(ws + wl.wsr_sync) + SYNC_FFLAG;   // [8][512] chunk flags
    unsigned* cnt = reinterpret_cast<unsigned*>(ws + wl.wsr_sync) + SYNC_FCNT;      // [8] arrival counter lines
    unsigned* gerr = (st.rt && st.rt->wsr_err_dev) ? st.rt->wsr_err_dev + 2 : nullptr;
    for (int l = 0; l < L; ++l) ED_CHECK_ARG(g[l].nchunks <= 512, "encoder_stack: too many chunks for the flag table");
    // (flags and counters were zeroed by the caller before the streams forked)
    // TWO side streams: what follows a full-rate layer (its LayerNorm, the next layer's product) runs on the
    // chunk-GEMM stream, what follows a layer behind the time reduction on the CALLER's stream - idle until the
    // join and already one of the four hardware queues.  One side stream was the bottleneck: per recurrence
    // launch (55 us) it had an event wait, the norm launch (12-17 us beside a full chip), 1-2 products
    // (20-55 us) and their flag kernels to carry.  Layer 0's products need nothing from the recurrence: all of
    // them go onto the caller's stream up front, behind the input LayerNorm that feeds them.
    int split = L;
    for (int l = 0; l < L; ++l)
        if (d->layers[l].reduce == 2) { split = l + 1; break; }
    if (split >= L) split = (L + 1) / 2;
    auto side = [&](int l) -> hipStream_t { return (st.serial || l < split) ? st.S[0] : st.C; };
    const int WGS = (H >> 4) * ((B + 63) >> 6);
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        return n;
    }();
    const int max_slots = max(1, min(ED_STACK_MAX_SLOTS, (g_trace ? 256 : n_cu) / WGS));
    const char* e_m = getenv("EDGEDICT_LPW_MARGIN");
    const int margin = (e_m && atoi(e_m) > 0) ? atoi(e_m) : 2;

    std::vector<std::vector<hipEvent_t>> Eg(L);
    std::vector<std::vector<char>> queued(L);
    for (int l = 0; l < L; ++l) {
        Eg[l].assign(g[l].nchunks, nullptr);
        queued[l].assign(g[l].nchunks, 0);
    }
    int next_g0 = 0;
    auto feed_layer0 = [&](int upto) -> int {
        for (; next_g0 < g[0].nchunks && next_g0 <= upto; ++next_g0) {
            ED_DEV(input_gemm(d, g, 0, next_g0, st.C));
            if (g_trace) g_trace->chunk_enqueued[g_trace->coff[0] + next_g0] = g_trace->launches;
            if (soft) ED_DEV(ed_stack_set_flag(fflag + next_g0, st.C));
            else ED_TRY(st.record(Eg[0][next_g0], st.C));
            queued[0][next_g0] = 1;
        }
        return ED_OK;
    };
    ED_TRY(feed_layer0(g[0].nchunks));
    std::vector<int> next_t(L, 0);
    std::vector<std::vector<int>> ready_w(L);
    for (int l = 0; l < L; ++l) ready_w[l].assign(g[l].nchunks, l == 0 ? 0 : 0x3fffffff);
    int launches = 0, idle = 0;
    if (st.rt) st.rt->stamp_used[0] = 0;
    if (st.rt && st.rt->tev[0][0]) ED_CHECK_HIP(hipEventRecord(st.rt->tev[0][0], st.R));
    const int T_out = (g[L - 1].T + d->layers[L - 1].reduce - 1) / d->layers[L - 1].reduce;
    struct Ran { int l, t0, t1; };
    for (int w = 0;; ++w) {
        bool finished = true;
        for (int l = 0; l < L; ++l) finished = finished && next_t[l] >= g[l].T;
        if (finished) break;
        EdLpwLaunch Lc;
        Lc.nslot = 0;
        Lc.data_poll = lpw_data_poll() ? 1 : 0;
        Lc.B = B;
        Lc.H = H;
        Ran ran[ED_STACK_MAX_SLOTS];
        for (int l = 0; l < L; ++l) {
            const edgedict_stack_layer_t& y = d->layers[l];
            const int t = next_t[l];
            if (t >= g[l].T || Lc.nslot >= max_slots) continue;
            const int k = t / g[l].cf;
            const bool opens = (t % g[l].cf == 0);
            if (opens && l > 0 && !(queued[l][k] && w >= ready_w[l][k])) continue;
            int m_min = g[l].m;
            for (int j = 0; j < l; ++j)
                if (next_t[j] < g[j].T) m_min = min(m_min, g[j].m);
            if (!Pace::allows(w, l, g[l].m, m_min)) continue;
            if (opens) {
                if (l == 0) ED_TRY(feed_layer0(k + 2));
                ED_CHECK_ARG(queued[l][k], "encoder_stack: schedule violated (layer %d chunk %d)", l, k);
                if (!soft) ED_TRY(st.wait(st.R, Eg[l][k]));
            }
            // (never across a chunk boundary.  Letting a single left-over frame - T = 401 = 25 x 16 + 1 - ride along, as the
            // BPTT does, moves the parity of the drain and costs a launch here: 36 instead of 35 at E6D2, dry run)
            const int t1 = min(t + nsub, min(g[l].T, (k + 1) * g[l].cf));
            EdLpwSlot& sl = Lc.slot[Lc.nslot];
            sl.G = bptr(y.G) + (long long)t * B * 4 * H;
            sl.img = bptr(ws + wl.himg[l]);
            sl.img_stride = (long long)wl.himg_stride;
            sl.img_bytes = (long long)(y.T + 1) * (long long)wl.himg_stride;
            sl.Y = bptr(y.Yx) + (long long)(t + 1) * BH;
            sl.C_prev = y.Cx + (long long)t * BH;
            sl.C = y.Cx + (long long)(t + 1) * BH;
            sl.Wfrag = bptr(y.whh_f);
            sl.counter = cnt + (size_t)l * LPW_CNT_STRIDE;
            sl.base = (unsigned)WGS * (unsigned)t;
            sl.wait_flag = (soft && opens) ? fflag + l * 512 + k : nullptr;
            sl.t0 = t;
            sl.nsteps = t1 - t;
            sl.layer = l;
            ran[Lc.nslot].l = l;
            ran[Lc.nslot].t0 = t;
            ran[Lc.nslot].t1 = t1;
            ++Lc.nslot;
            next_t[l] = t1;
            if (g_trace)
                for (int tt = t; tt < t1; ++tt) g_trace->step_launch[g_trace->toff[l] + tt] = g_trace->launches;
        }
        if (Lc.nslot == 0) {
            ED_CHECK_ARG(++idle < 4096, "encoder_stack: forward schedule made no progress");
            continue;
        }
        idle = 0;
        ++launches;
        Lc.stamp = st.rt ? st.rt->stamp_slot(0, st.R) : nullptr;
        Lc.err = gerr;
        Lc.trace = g_wsr_trace;      // debug buffer registered with edgedict_stack_wsr_set_trace (normally null)
        ED_DEV(ed_stack_launch_fwd_lpw(Lc, st.R));
        if (g_trace) {
            g_trace->max_slots = max(g_trace->max_slots, Lc.nslot);
            ++g_trace->launches;
        }
        // ---- side streams: LayerNorm of what this launch finished, then the products of completed chunks
        for (int pass = 0; pass < 2; ++pass) {
            hipStream_t S = pass == 0 ? st.S[0] : st.C;
            EdChunkNorm items[ED_STACK_MAX_SLOTS];
            int idx[ED_STACK_MAX_SLOTS], ni = 0;
            for (int i = 0; i < Lc.nslot; ++i) {
                const int l = ran[i].l;
                if (side(l) != S) continue;
                const edgedict_stack_layer_t& y = d->layers[l];
                EdChunkNorm& e = items[ni];
                e.Yx1 = bptr(y.Yx) + BH;
                e.X = y.residual ? bptr(y.X) : nullptr;
                e.gamma = y.ln_gamma;
                e.beta = y.ln_beta;
                if (l + 1 < L) {
                    e.out = bptr(d->layers[l + 1].X);
                    e.out_st = BH;
                    e.out_sb = H;
                } else {
                    e.out = bptr(d->out);
                    e.out_st = H;
                    e.out_sb = (long long)T_out * H;
                }
                e.mean = y.mean;
                e.rstd = y.rstd;
                e.T = y.T;
                e.t0 = ran[i].t0;
                e.t1 = ran[i].t1;
                e.reduce = y.reduce;
                const DropCfg dc = drop_of(y);
                e.drop_thresh = dc.thresh; e.drop_seed = dc.seed; e.drop_scale = dc.scale; e.drop_T = dc.T_out;
                idx[ni++] = i;
            }
            if (ni == 0) continue;
            if (soft) {
                // order S behind this launch's steps through their arrival counters, not through an event
                const unsigned* cp[ED_STACK_MAX_SLOTS];
                unsigned tg[ED_STACK_MAX_SLOTS];
                for (int j = 0; j < ni; ++j) {
                    cp[j] = cnt + (size_t)ran[idx[j]].l * LPW_CNT_STRIDE;
                    tg[j] = (unsigned)WGS * (unsigned)ran[idx[j]].t1;
                }
                ED_DEV(ed_stack_wait_counters(cp, tg, ni, gerr, S));
            } else {
                ED_TRY(st.chain(st.R, S));
            }
            ED_DEV(ed_stack_multi_norm(items, ni, B, H, d->eps, S));
            for (int j = 0; j < ni; ++j) {
                const int i = idx[j];
                const int l = ran[i].l, k = ran[i].t0 / g[l].cf;
                if (l + 1 >= L || ran[i].t1 != min(g[l].T, (k + 1) * g[l].cf)) continue;
                ED_DEV(input_gemm(d, g, l + 1, k, S));
                if (g_trace) g_trace->chunk_enqueued[g_trace->coff[l + 1] + k] = g_trace->launches;
                if (soft) ED_DEV(ed_stack_set_flag(fflag + (l + 1) * 512 + k, S));
                else ED_TRY(st.record(Eg[l + 1][k], S));
                queued[l + 1][k] = 1;
                ready_w[l + 1][k] = w + margin;
            }
            if (st.serial) break;     // one stream: the first pass took every slot
        }
    }
    if (st.rt && st.rt->tev[0][1]) {
        ED_CHECK_HIP(hipEventRecord(st.rt->tev[0][1], st.R));
        st.rt->tlaunches[0] = launches;
    }
    ED_TRY(st.chain(st.R, st.C));
    for (int l = 0; l < L; ++l) ED_TRY(st.chain(st.S[l], st.C));
    return ED_OK;
}

// steps per launch of the launch-persistent forward for this geometry (0 = use the launch-per-step kernels):
// a divisor of the chunk, even when a layer halves the frame rate (its LayerNorm pairs frames)
int lpw_steps(const edgedict_stack_desc_t* d) {
    // read on every call (two getenv per forward pass): tests switch the path inside one process
    const char* e_on = getenv("EDGEDICT_STACK_LPW");
    const char* e_n = getenv("EDGEDICT_LPW_STEPS");
    const int on = e_on ? atoi(e_on) : 1, want = e_n ? atoi(e_n) : min(d->chunk, 16);      // default: a chunk per launch
    if (!on || !ed_stack_lpw_supported(d->B, d->H)) return 0;
    // the kernel addresses a layer's images through ONE 32-bit buffer descriptor, and all workgroups of a layer must be
    // resident at once (they wait for each other inside the launch): otherwise the launch-per-step kernels
    {
        const size_t B16 = (size_t)(d->B + 15) / 16 * 16;
        const size_t stride = align256(B16 * d->H * sizeof(bf16_t));
        for (int l = 0; l < d->L; ++l)
            if ((size_t)(d->layers[l].T + 1) * stride >= (1ull << 32)) return 0;
        if (!g_trace && device_cus() < (d->H >> 4) * ((d->B + 63) >> 6)) return 0;
    }
    bool reduces = false;
    for (int l = 0; l < d->L; ++l) reduces = reduces || d->layers[l].reduce == 2;
    int ns = max(1, min(want, d->chunk));
    while (ns > 1 && (d->chunk % ns != 0 || (reduces && (ns & 1)))) --ns;
    if (reduces && (ns & 1)) return 0;
    return ns;
}

// split-K weights-stationary BPTT (stack_bwd_sk_kernel): steps per launch, 0 = not applicable / switched off
int sk_bwd_steps(const edgedict_stack_desc_t* d) {
    const char* e_on = getenv("EDGEDICT_STACK_BWD_SK");
    const char* e_n = getenv("EDGEDICT_SK_STEPS");
    const int on = e_on ? atoi(e_on) : 1, want = e_n ? atoi(e_n) : min(d->chunk, 16);      // default: a chunk per launch
    if (!on || !ed_stack_sk_supported(d->B, d->H)) return 0;
    for (int l = 0; l < d->L; ++l)
        if (!d->layers[l].whh_s) return 0;
    {   // as lpw_steps: 32-bit image range, and the layer's (H / 64) x 4 workgroups co-resident (two fit a CU)
        const size_t B16 = (size_t)(d->B + 15) / 16 * 16;
        const size_t stride = align256(B16 * 4 * d->H * sizeof(bf16_t));
        for (int l = 0; l < d->L; ++l)
            if ((size_t)(d->layers[l].T + 1) * stride >= (1ull << 32)) return 0;
        if (!g_trace && 2 * device_cus() < (d->H >> 6) * 4) return 0;
    }
    int ns = max(1, min(want, d->chunk));
    while (ns > 1 && d->chunk % ns != 0) --ns;
    return ns;
}

}  // namespace

extern "C" int edgedict_stack_forward(const edgedict_stack_desc_t* d, void* stream_) {
    std::vector<Geom> g;
    ED_TRY(validate(d, g, false));
    const int B = d->B, H = d->H, L = d->L;
    const long long BH = (long long)B * H;
    const WsLayout wl = ws_layout(d);
    char* ws = (char*)d->ws;
    Streams st;
    ED_TRY(open_streams(d, stream_, st));
    // TWO recurrence streams: the layers behind the time reduction run their launches on the CALLER's stream,
    // which is otherwise idle until this call's join - the 3.4 us between two dependent launches of one
    // stream (12.5 us kernel, 15.9 us period) are then covered by the other stream's kernel.  Per-layer state
    // stays on one stream, layers meet through the chunk events only.  Measured: forward 7.73 -> 7.18 ms,
    // step 26.5 -> 25.9 ms (EDGEDICT_STACK_FWD_R2=0 for one stream).  The auxiliary stream is NOT a
    // candidate (the prediction network runs there: 7.20 ms forward but a slower step), a fifth stream
    // neither (hardware queues), and splitting the layers by parity instead of by rate is slower
    // (7.4 ms) and lets adjacent layers of one launch race.
    static const int fwd_r2 = [] { const char* e = getenv("EDGEDICT_STACK_FWD_R2"); return e ? atoi(e) : 1; }();
    if (fwd_r2 && !st.serial && st.R2 == st.R) st.R2 = st.C;
    st.split = L;
    if (st.R2 != st.R) {   // full-rate layers (up to the first time reduction) vs the rest
        st.split = L / 2;
        for (int l = 0; l < L; ++l)
            if (d->layers[l].reduce == 2) { st.split = l + 1; break; }
        if (st.split >= L) st.split = L / 2;
    }

    // ---- prologue on the caller's stream: input LayerNorm (-> X_0, time-major), initial states
    ED_DEV(ed_stack_input_norm(d->x_dtype, d->x, d->in_gamma, d->in_beta, bptr(d->layers[0].X),
                               d->in_mean, d->in_rstd, B, d->T0, d->I0, d->eps, st.C));
    const int lpw_ns = lpw_steps(d);
    // test aid (EDGEDICT_STACK_POISON=1): fill the per-frame h images with NaN patterns before the pass - the
    // workspace usually still holds the images of the pass before, which on identical inputs are the RIGHT values, so
    // a read that overtakes its producer would go unnoticed; with the poison it turns every result into NaN
    // ... and it is how the data-polling forward kernel works at all (stack_fwd_lpw_kernel<DP>): readers recognise a
    // chunk that has not been written yet by this pattern (205 MB for E6D2 at 15 s: ~50 us of memset)
    // (the layers' images are contiguous in the workspace: one memset)
    if (lpw_ns && (poison_on() || lpw_data_poll()))
        ED_DEV(ed_stack_fill(ws + wl.himg[0], 0xff,
                             wl.himg[L - 1] + (size_t)(d->layers[L - 1].T + 1) * wl.himg_stride - wl.himg[0], st.C));
    if (st.rt) {
        st.rt->tkind[0] = lpw_ns ? 1 : 0;
        st.rt->tsteps[0] = lpw_ns;
    }
    for (int l0 = 0; l0 < L; l0 += 8) {     // initial states, eight layers per launch
        EdInitStates A;
        A.n = min(8, L - l0);
        A.B = B;
        A.H = H;
        for (int j = 0; j < A.n; ++j) {
            const int l = l0 + j;
            const edgedict_stack_layer_t& y = d->layers[l];
            A.h0[j] = d->h0 ? d->h0 + l * BH : nullptr;
            A.c0[j] = d->c0 ? d->c0 + l * BH : nullptr;
            A.Yx0[j] = bptr(y.Yx);
            A.Cx0[j] = y.Cx;
            A.hfrag[j] = bptr(ws + (lpw_ns ? wl.himg[l] : wl.frag0[l]));
        }
        ED_DEV(ed_stack_init_states(A, st.C));
    }
    // chunk flags and arrival counters of the launch-persistent pass: one memset over the head of the sync region (it
    // covers the backward pass's flags too, which that pass zeroes again), before the streams fork
    if (lpw_ns) ED_DEV(ed_stack_zero(ws + wl.wsr_sync, SYNC_BCNT * sizeof(unsigned), st.C));
    ED_TRY(st.chain(st.C, st.R));
    if (st.R2 != st.R) ED_TRY(st.chain(st.C, st.R2));
    for (int l = 0; l < L; ++l) ED_TRY(st.chain(st.C, st.S[l]));

    // flag waits instead of stream waits on the recurrence streams (stack_kernels.hip soft_wait): one word per
    // (layer, chunk) in the workspace's sync region, zeroed before the streams fork
    static const int soft_env = [] { const char* e = getenv("EDGEDICT_STACK_SOFT_WAIT"); return e ? atoi(e) : 1; }();
    const bool soft = soft_env && !st.serial && L <= 8;
    if (lpw_ns) return forward_lpw(d, g, st, wl, lpw_ns, soft);
    unsigned* fflag = reinterpret_cast<unsigned*>(ws + wl.wsr_sync) + SYNC_FFLAG;           // [8][512]
    unsigned* gerr = (st.rt && st.rt->wsr_err_dev) ? st.rt->wsr_err_dev + 2 : nullptr;
    if (soft) {
        for (int l = 0; l < L; ++l) ED_CHECK_ARG(g[l].nchunks <= 512, "encoder_stack: too many chunks for the flag table");
        ED_DEV(ed_stack_zero(fflag, (size_t)8 * 512 * sizeof(unsigned), st.C));
        ED_TRY(st.chain(st.C, st.R));
        if (st.R2 != st.R) ED_TRY(st.chain(st.C, st.R2));
        for (int l = 0; l < L; ++l) ED_TRY(st.chain(st.C, st.S[l]));
    }
    std::vector<std::vector<hipEvent_t>> Eg(L);
    std::vector<std::vector<char>> queued(L);   // schedule self-check: producer enqueued before consumer
    for (int l = 0; l < L; ++l) {
        Eg[l].assign(g[l].nchunks, nullptr);
        queued[l].assign(g[l].nchunks, 0);
    }
    int next_g0 = 0;   // next chunk of layer 0 whose input product has not been enqueued
    auto feed_layer0 = [&](int upto) -> int {
        for (; next_g0 < g[0].nchunks && next_g0 <= upto; ++next_g0) {
            ED_DEV(input_gemm(d, g, 0, next_g0, st.S[0]));
            if (g_trace) g_trace->chunk_enqueued[g_trace->coff[0] + next_g0] = g_trace->launches;
            if (soft) ED_DEV(ed_stack_set_flag(fflag + next_g0, st.S[0]));
            else ED_TRY(st.record(Eg[0][next_g0], st.S[0]));
            queued[0][next_g0] = 1;
        }
        return ED_OK;
    };
    ED_TRY(feed_layer0(1));

    const int margin = margin_launches(d, g, false);
    std::vector<int> next_t(L, 0), stepped_w(L, -2);
    std::vector<std::vector<int>> ready_w(L);
    for (int l = 0; l < L; ++l) ready_w[l].assign(g[l].nchunks, l == 0 ? 0 : 0x3fffffff);
    int launches = 0, idle = 0;
    if (st.rt) st.rt->stamp_used[0] = 0;
    if (st.rt && st.rt->tev[0][0]) ED_CHECK_HIP(hipEventRecord(st.rt->tev[0][0], st.R));
    const int T_out = (g[L - 1].T + d->layers[L - 1].reduce - 1) / d->layers[L - 1].reduce;
    struct Done { int l, k; };
    for (int w = 0;; ++w) {
        bool finished = true;
        for (int l = 0; l < L; ++l) finished = finished && next_t[l] >= g[l].T && stepped_w[l] < w - 1;
        if (finished) break;
        EdFwdLaunch Lcs[2];
        for (auto& x : Lcs) {
            x.nstep = x.nnorm = 0;
            x.B = B; x.H = H; x.eps = d->eps;
        }
        Done done[ED_STACK_MAX_SLOTS];
        int ndone = 0;
        for (int l = 0; l < L; ++l) {
            const edgedict_stack_layer_t& y = d->layers[l];
            EdFwdLaunch& Lc = Lcs[st.which(l)];
            // ---- LayerNorm of the frame the previous launch finished (before the step below
            // advances next_t)
            if (stepped_w[l] == w - 1) {
                const int t = next_t[l] - 1;
                int ta = -1, tb = -1;
                if (y.reduce == 1) ta = t;
                else if (t & 1) { ta = t - 1; tb = t; }
                else if (t == g[l].T - 1) ta = t;
                if (ta >= 0) {
                    const int tau = ta / y.reduce;
                    EdFwdNorm& n = Lc.norm[Lc.nnorm++];
                    n.y0 = bptr(y.Yx) + (long long)(ta + 1) * BH;
                    n.r0 = y.residual ? bptr(y.X) + (long long)ta * BH : nullptr;
                    n.y1 = tb >= 0 ? bptr(y.Yx) + (long long)(tb + 1) * BH : nullptr;
                    n.r1 = (tb >= 0 && y.residual) ? bptr(y.X) + (long long)tb * BH : nullptr;
                    n.gamma = y.ln_gamma;
                    n.beta = y.ln_beta;
                    if (l + 1 < L) {
                        n.out = bptr(d->layers[l + 1].X) + (long long)tau * BH;
                        n.out_stride = H;
                    } else {
                        n.out = bptr(d->out) + (long long)tau * H;
                        n.out_stride = (long long)T_out * H;
                    }
                    n.mean0 = y.mean + (long long)ta * B;
                    n.rstd0 = y.rstd + (long long)ta * B;
                    n.mean1 = tb >= 0 ? y.mean + (long long)tb * B : nullptr;
                    n.rstd1 = tb >= 0 ? y.rstd + (long long)tb * B : nullptr;
                    n.scale = y.reduce == 1 ? 1.f : 0.5f;
                    const DropCfg dc = drop_of(y);
                    n.drop_thresh = dc.thresh; n.drop_seed = dc.seed; n.drop_scale = dc.scale; n.drop_T = dc.T_out;
                    n.tau = tau;
                }
                // last frame of a chunk: the next layer's input rows of this chunk are complete
                const int k = t / g[l].cf;
                if (l + 1 < L && t == min(g[l].T, (k + 1) * g[l].cf) - 1) {
                    done[ndone].l = l;
                    done[ndone].k = k;
                    ++ndone;
                }
            }
            // ---- time step of layer l
            const int t = next_t[l];
            if (t >= g[l].T) continue;
            const int k = t / g[l].cf;
            const bool opens = (t % g[l].cf == 0);
            if (opens && !(queued[l][k] && w >= ready_w[l][k]) && l > 0) continue;
            int m_min = g[l].m;   // slowest-changing layers are paced by the faster ones still running
            for (int j = 0; j < l; ++j)
                if (next_t[j] < g[j].T) m_min = min(m_min, g[j].m);
            if (!Pace::allows(w, l, g[l].m, m_min)) continue;
            if (opens) {
                if (l == 0) ED_TRY(feed_layer0(k + 2));
                ED_CHECK_ARG(queued[l][k], "encoder_stack: schedule violated (layer %d chunk %d)", l, k);
                if (!soft) ED_TRY(st.wait(st.RS(l), Eg[l][k]));
            }
            EdFwdStep& sl = Lc.step[Lc.nstep++];
            sl.wait_flag = (soft && opens) ? fflag + l * 512 + k : nullptr;
            bf16_t* f0 = bptr(ws + wl.frag0[l]);
            bf16_t* f1 = bptr(ws + wl.frag1[l]);
            sl.G_t = bptr(y.G) + (long long)t * B * 4 * H;
            sl.hfrag_in = (t & 1) ? f1 : f0;
            sl.hfrag_out = (t & 1) ? f0 : f1;
            sl.Y_t = bptr(y.Yx) + (long long)(t + 1) * BH;
            sl.C_prev = y.Cx + (long long)t * BH;
            sl.C_t = y.Cx + (long long)(t + 1) * BH;
            sl.Wfrag = bptr(y.whh_f);
            stepped_w[l] = w;
            next_t[l] = t + 1;
            if (g_trace) g_trace->step_launch[g_trace->toff[l] + t] = g_trace->launches;
        }
        if (Lcs[0].nstep + Lcs[0].nnorm + Lcs[1].nstep + Lcs[1].nnorm == 0) {
            // every unfinished layer waits for a side-stream product: nothing to overlap it with
            ED_CHECK_ARG(++idle < 4096, "encoder_stack: forward schedule made no progress");
            continue;
        }
        idle = 0;
        ++launches;
        Lcs[0].stamp = st.rt ? st.rt->stamp_slot(0, st.R) : nullptr;
        Lcs[1].stamp = nullptr;
        Lcs[0].err = Lcs[1].err = gerr;
        ED_DEV(ed_stack_launch_fwd(Lcs[0], st.R));
        if (st.R2 != st.R) ED_DEV(ed_stack_launch_fwd(Lcs[1], st.R2));
        if (g_trace) {
            g_trace->max_slots = max(g_trace->max_slots, max(Lcs[0].nstep + Lcs[1].nstep, Lcs[0].nnorm + Lcs[1].nnorm));
            ++g_trace->launches;
        }
        for (int i = 0; i < ndone; ++i) {
            const int l = done[i].l + 1, k = done[i].k;
            ED_TRY(st.chain(st.RS(done[i].l), st.S[l]));
            ED_DEV(input_gemm(d, g, l, k, st.S[l]));
            if (g_trace) g_trace->chunk_enqueued[g_trace->coff[l] + k] = g_trace->launches;
            if (soft) ED_DEV(ed_stack_set_flag(fflag + l * 512 + k, st.S[l]));
            else ED_TRY(st.record(Eg[l][k], st.S[l]));
            queued[l][k] = 1;
            ready_w[l][k] = w + margin;
        }
    }
    const int Wtot = launches;
    if (st.rt && st.rt->tev[0][1]) {
        ED_CHECK_HIP(hipEventRecord(st.rt->tev[0][1], st.R));
        st.rt->tlaunches[0] = Wtot;
    }
    ED_TRY(st.chain(st.R, st.C));
    if (st.R2 != st.R) ED_TRY(st.chain(st.R2, st.C));
    for (int l = 0; l < L; ++l) ED_TRY(st.chain(st.S[l], st.C));
    return ED_OK;
}

extern "C" int edgedict_stack_pack_sk(const float* w_hh, int H, void* whh_s, void* stream_) {
    ED_CHECK_ARG(H >= 64 && H % 64 == 0 && H <= 1024, "stack_pack_sk: needs H %% 64 == 0 and H <= 1024 (got %d)", H);
    ED_CHECK_ARG(w_hh && whh_s, "stack_pack_sk: null pointer");
    return ed_stack_pack_sk(w_hh, (bf16_t*)whh_s, H, (hipStream_t)stream_);
}

extern "C" int edgedict_stack_wsr_set_trace(void* device_buffer) {
    g_wsr_trace = (long long*)device_buffer;
    return ED_OK;
}

extern "C" int edgedict_stack_wsr_error(void) {
    Runtime* r = runtime_for_current_device();
    std::unique_lock<std::mutex> lock;
    if (r) lock = std::unique_lock<std::mutex>(r->mu);
    if (!r || !r->wsr_err_host) return 0;
    volatile unsigned* p = r->wsr_err_host;
    if (!(p[0] | p[1] | p[2])) return 0;         // the normal case: nothing is written, nothing races
    // A give-up code.  The words are also what the guarded Adam step of the SAME training step reads on the device
    // (edgedict_adam_step_guarded) - and the host runs ahead of the device: clearing them now could hide the failure
    // from a guard kernel that is enqueued but has not run yet, and the garbage gradients would be applied.  So the
    // device drains first (an error path: the caller is about to raise), then the words are cleared
    (void)hipDeviceSynchronize();
    const unsigned code = p[2] ? p[2] : (p[1] ? p[1] : p[0]);   // [2]: flag waits of the step kernels
    p[0] = 0;
    p[1] = 0;
    p[2] = 0;
    return (int)code;
}

extern "C" void* edgedict_stack_error_words(int host) {
    Runtime* r = runtime_for_current_device();
    std::unique_lock<std::mutex> lock;
    if (r) lock = std::unique_lock<std::mutex>(r->mu);
    if (!r || !r->wsr_err_host || !r->wsr_err_dev) {
        ed_set_error("stack_error_words: no device runtime");
        return nullptr;
    }
    return host ? (void*)r->wsr_err_host : (void*)r->wsr_err_dev;
}

extern "C" int edgedict_stack_schedule(const edgedict_stack_desc_t* d, int backward, int32_t* step_launch,
                                       int32_t* chunk_enqueued, int32_t* n_launches, int32_t* max_slots) {
    ED_CHECK_ARG(d && step_launch && chunk_enqueued && n_launches && max_slots, "stack_schedule: null pointer");
    ScheduleTrace tr;
    {
        std::vector<Geom> g;
        ED_TRY(validate(d, g, backward != 0));
        int to = 0, co = 0;
        for (int l = 0; l < d->L; ++l) {
            tr.toff.push_back(to);
            tr.coff.push_back(co);
            to += g[l].T;
            co += g[l].nchunks;
        }
        for (int i = 0; i < to; ++i) step_launch[i] = -1;
        for (int i = 0; i < co; ++i) chunk_enqueued[i] = -1;
    }
    tr.step_launch = step_launch;
    tr.chunk_enqueued = chunk_enqueued;
    g_trace = &tr;
    const int rc = backward ? edgedict_stack_backward(d, nullptr) : edgedict_stack_forward(d, nullptr);
    g_trace = nullptr;
    *n_launches = tr.launches;
    *max_slots = tr.max_slots;
    return rc;
}

extern "C" int edgedict_stack_last_timing(int backward, float* ms, int* launches) {
    ED_CHECK_ARG(ms && launches, "stack_last_timing: null pointer");
    Runtime* r = runtime_for_current_device();
    std::unique_lock<std::mutex> lock;
    if (r) lock = std::unique_lock<std::mutex>(r->mu);
    const int i = backward ? 1 : 0;
    ED_CHECK_ARG(r && r->tev[i][0] && r->tev[i][1] && r->tlaunches[i] > 0, "stack_last_timing: nothing recorded");
    ED_CHECK_HIP(hipEventSynchronize(r->tev[i][1]));
    ED_CHECK_HIP(hipEventElapsedTime(ms, r->tev[i][0], r->tev[i][1]));
    *launches = r->tlaunches[i];
    return ED_OK;
}

extern "C" int edgedict_stack_last_mode(int backward, int* kind, int* steps_per_launch) {
    ED_CHECK_ARG(kind && steps_per_launch, "stack_last_mode: null pointer");
    Runtime* r = runtime_for_current_device();
    std::unique_lock<std::mutex> lock;
    if (r) lock = std::unique_lock<std::mutex>(r->mu);
    ED_CHECK_ARG(r, "stack_last_mode: no device runtime");
    *kind = r->tkind[backward ? 1 : 0];
    *steps_per_launch = r->tsteps[backward ? 1 : 0];
    return ED_OK;
}

extern "C" int edgedict_stack_time_launches(int on) {
    Runtime* r = runtime_for_current_device();
    std::unique_lock<std::mutex> lock;
    if (r) lock = std::unique_lock<std::mutex>(r->mu);
    ED_CHECK_ARG(r, "stack_time_launches: no device runtime");
    r->time_each = on != 0;
    return ED_OK;
}

extern "C" int edgedict_stack_launch_times(int backward, float* sum_ms, int* launches) {
    ED_CHECK_ARG(sum_ms && launches, "stack_launch_times: null pointer");
    Runtime* r = runtime_for_current_device();
    std::unique_lock<std::mutex> lock;
    if (r) lock = std::unique_lock<std::mutex>(r->mu);
    const int i = backward ? 1 : 0;
    ED_CHECK_ARG(r && r->stamps[i] && r->stamp_used[i] > 0,
                 "stack_launch_times: nothing recorded (edgedict_stack_time_launches(1) first)");
    const int n = r->stamp_used[i];
    std::vector<unsigned long long> h(2 * (size_t)n);
    ED_CHECK_HIP(hipStreamSynchronize(r->R));
    ED_CHECK_HIP(hipMemcpy(h.data(), r->stamps[i], h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double ticks = 0.0;
    int ok = 0;
    for (int k = 0; k < n; ++k)
        if (h[2 * k + 1] > h[2 * k]) {
            ticks += (double)(h[2 * k + 1] - h[2 * k]);
            ++ok;
        }
    ED_CHECK_ARG(ok > 0, "stack_launch_times: no launch stamped");
    *sum_ms = (float)(ticks * 1e-5);      // 100 MHz ticks -> ms
    *launches = ok;
    return ED_OK;
}

extern "C" int edgedict_stack_launch_stamps(int backward, unsigned long long* out, int max_launches, int* launches) {
    ED_CHECK_ARG(out && launches && max_launches > 0, "stack_launch_stamps: bad arguments");
    Runtime* r = runtime_for_current_device();
    std::unique_lock<std::mutex> lock;
    if (r) lock = std::unique_lock<std::mutex>(r->mu);
    const int i = backward ? 1 : 0;
    ED_CHECK_ARG(r && r->stamps[i] && r->stamp_used[i] > 0,
                 "stack_launch_stamps: nothing recorded (edgedict_stack_time_launches(1) first)");
    const int n = min(max_launches, r->stamp_used[i]);
    ED_CHECK_HIP(hipDeviceSynchronize());
    ED_CHECK_HIP(hipMemcpy(out, r->stamps[i], (size_t)n * 16, hipMemcpyDeviceToHost));
    *launches = n;
    return ED_OK;
}

extern "C" int edgedict_stack_backward(const edgedict_stack_desc_t* d, void* stream_) {
    std::vector<Geom> g;
    ED_TRY(validate(d, g, true));
    ED_CHECK_ARG(d->dout && d->d_in_gamma && d->d_in_beta, "encoder_stack: null backward pointer in descriptor");
    ED_CHECK_ARG(!(d->flags & EDGEDICT_STACK_INFERENCE), "encoder_stack: backward on a descriptor flagged EDGEDICT_STACK_INFERENCE");
    const int B = d->B, H = d->H, L = d->L;
    const long long BH = (long long)B * H;
    const WsLayout wl = ws_layout(d);
    char* ws = (char*)d->ws;
    Streams st;
    ED_TRY(open_streams(d, stream_, st));
    // ONE recurrence stream in the backward pass (a second one was within noise for the launch-per-step kernels and
    // the macro-step kernels enqueue every launch on st.R: a layer's side work must be ordered behind that stream)
    st.R2 = st.R;
    st.split = L;
    const int T_out = (g[L - 1].T + d->layers[L - 1].reduce - 1) / d->layers[L - 1].reduce;

    static const int soft_env = [] { const char* e = getenv("EDGEDICT_STACK_SOFT_WAIT"); return e ? atoi(e) : 1; }();
    const bool soft = soft_env && !st.serial && L <= 8;       // flag waits, as in the forward pass
    unsigned* bflag = reinterpret_cast<unsigned*>(ws + wl.wsr_sync) + SYNC_BFLAG;   // [8][512], after the forward's
    unsigned* gerr = (st.rt && st.rt->wsr_err_dev) ? st.rt->wsr_err_dev + 2 : nullptr;
    if (soft)
        for (int l = 0; l < L; ++l) ED_CHECK_ARG(g[l].nchunks <= 512, "encoder_stack: too many chunks for the flag table");
    // split-K weights-stationary BPTT (stack_bwd_sk_kernel): steps per launch (0 = one launch per step) and the
    // layers' arrival counters, behind the forward pass's in the sync region
    const int sk_ns = sk_bwd_steps(d);
    if (st.rt) {
        st.rt->tkind[1] = sk_ns ? 2 : 0;
        st.rt->tsteps[1] = sk_ns;
    }
    unsigned* cntb = reinterpret_cast<unsigned*>(ws + wl.wsr_sync) + SYNC_BCNT;
    unsigned* gcnt = reinterpret_cast<unsigned*>(ws + wl.wsr_sync) + SYNC_GCNT;     // 64 KB into the sync region: [8][SK_CNT_LINES][64]
    if (sk_ns) {
        if (poison_on())        // as in the forward pass: dG images and split-K partials of the pass before -> NaN
            for (int l = 0; l < L; ++l) {
                ED_DEV(ed_stack_fill(ws + wl.gimg[l], 0xff, (size_t)(d->layers[l].T + 1) * wl.gimg_stride, st.C));
                ED_DEV(ed_stack_fill(ws + wl.skpart[l], 0xff, (size_t)2 * (H / 64) * 4 * 64 * 64 * sizeof(float), st.C));
            }
    }
    // ---- prologue: chunk flags + arrival counters (everything in the sync region behind the forward pass's flags,
    // one memset), running dL/dc = 0 (contiguous, one memset)
    if (soft || sk_ns)
        ED_DEV(ed_stack_zero(ws + wl.wsr_sync + SYNC_BFLAG * sizeof(unsigned), WSR_SYNC_BYTES - SYNC_BFLAG * sizeof(unsigned), st.C));
    ED_DEV(ed_stack_zero(ws + wl.dC[0], wl.dC[L - 1] + (size_t)BH * sizeof(float) - wl.dC[0], st.C));
    ED_TRY(st.chain(st.C, st.R));
    if (st.R2 != st.R) ED_TRY(st.chain(st.C, st.R2));
    for (int l = 0; l < L; ++l) ED_TRY(st.chain(st.C, st.S[l]));
    ED_TRY(st.chain(st.C, st.W));

    std::vector<std::vector<hipEvent_t>> Eb(L);
    std::vector<std::vector<char>> queued(L);
    for (int l = 0; l < L; ++l) {
        Eb[l].assign(g[l].nchunks, nullptr);
        queued[l].assign(g[l].nchunks, l == L - 1 ? 1 : 0);
    }
    // ---- LayerNorm backward of the TOP layer, chunk by chunk on the side stream, last chunk first: the
    // BPTT starts after ONE chunk's kernel (~10 us) instead of after all frames' (150 us on the caller's
    // stream, with nothing beside it)
    {
        const edgedict_stack_layer_t& y = d->layers[L - 1];
        hipStream_t S = st.S[L - 1];
        const DropCfg dcT = drop_of(y);
        for (int k = g[L - 1].nchunks - 1; k >= 0; --k) {
            const int t0 = k * g[L - 1].cf, t1 = min(y.T, t0 + g[L - 1].cf);
            ED_DEV(ed_stack_ln_bwd(bptr(d->dout), H, (long long)T_out * H, bptr(y.Yx) + BH,
                                   y.residual ? bptr(y.X) : nullptr, y.ln_gamma, y.mean, y.rstd, bptr(y.dZ),
                                   (float*)(ws + wl.lnpart[L - 1]) + (size_t)k * LNB_GRID * 2 * H, LNB_GRID, B, H,
                                   t0, t1, y.reduce, S, dcT.thresh, dcT.seed, dcT.scale, dcT.T_out));
            if (soft) ED_DEV(ed_stack_set_flag(bflag + (L - 1) * 512 + k, S));
            else ED_TRY(st.record(Eb[L - 1][k], S));
        }
    }
    std::vector<int> deferred;   // layers whose weight gradients run after the BPTT (DW_AT_END)

    const int acc_grads = (d->flags & EDGEDICT_STACK_ACCUM_GRADS) ? 1 : 0;
    // weight gradients of layer l over frames [t0, t1): dW_ih (+)= dG^T X, dW_hh (+)= dG^T H_prev,
    // db (+)= colsum dG.  Issued per SEGMENT of `dw_seg` chunks as the layer's BPTT passes them, so
    // the products are spread under the whole recurrence instead of piling up behind the last
    // layers (the BPTT launches next to a busy W stream take 27 us instead of 16).
    auto weight_grads = [&](int l, int t0, int t1, bool first, bool last_of_all) -> int {
        if (g_trace) {
            // dry run: no product, but the host callback fires at the same point of the schedule - a data-parallel
            // host can be tested for WHEN each layer's bucket may leave (tests/test_dp_gloo.py: ranks whose batches
            // differ in length run different launch counts and must still issue the same collectives in the same order)
            if (t0 == 0 && d->grads_final) d->grads_final(l, d->grads_final_user);
            return ED_OK;
        }
        const edgedict_stack_layer_t& y = d->layers[l];
        const int M = (t1 - t0) * B;
        const long long r0 = (long long)t0 * B;
        float* tmpW = (float*)(ws + wl.tmpW);
        float* tmpB = (float*)(ws + wl.tmpB);
        // the very last product has nothing left to disturb: it may take the whole chip
        const int sk = d->split_k > 0 ? d->split_k : (last_of_all ? 8 : 2);
        const int cap = last_of_all ? 4 : 2;
        const int acc = (acc_grads || !first) ? 1 : 0;
        int S = 1;
        // QUIET products (K slices written once, no atomics: a concurrent kernel with dirty lines
        // makes every BPTT launch boundary 3-10x dearer); rows come out in interleaved gate order
        // and are summed + un-permuted in one pass
        ED_TRY(ed_gemm_quiet_partials(ED_BF16, bptr(y.G) + r0 * 4 * H, 4ll * H, 0, bptr(y.X) + r0 * y.I, y.I, 0,
                                      4 * H, y.I, M, sk, cap, tmpW, &S, st.W));
        ED_TRY(unpermute_rows(tmpW, 4ll * H * y.I, S, y.dW_ih, nullptr, H, y.I, acc, st.W));
        ED_TRY(ed_gemm_quiet_partials(ED_BF16, bptr(y.G) + r0 * 4 * H, 4ll * H, 0, bptr(y.Yx) + r0 * H, H, 0,
                                      4 * H, H, M, sk, cap, tmpW, &S, st.W));
        ED_TRY(unpermute_rows(tmpW, 4ll * H * H, S, y.dW_hh, nullptr, H, H, acc, st.W));
        ED_TRY(ed_stack_zero(tmpB, (size_t)4 * H * sizeof(float), st.W));
        ED_TRY(edgedict_colsum(ED_BF16, bptr(y.G) + r0 * 4 * H, 4ll * H, tmpB, M, 4 * H, st.W));
        hipLaunchKernelGGL(unpermute_rows_kernel, dim3(ed_grid_for(4ll * H, 256, 4096)), dim3(256),
                           0, st.W, tmpB, 0, 1, y.db, y.db_hh, H, 1, acc);
        ED_CHECK_LAUNCH("unpermute_rows_kernel");
        // frames [0, t1) were the last ones of this layer: its weight gradients are final once the
        // auxiliary stream reaches this point
        if (t0 == 0 && d->grads_final) d->grads_final(l, d->grads_final_user);
        return ED_OK;
    };
    int dw_seg = 1 << 20;   // measured: 8 / 4 / 2 / 1 chunks per segment cost +0.3 / +0.6 / +1.6 / +5.3 ms per step
    if (const char* e = getenv("EDGEDICT_STACK_DW_SEG")) dw_seg = max(1, atoi(e));
    if (d->flags & EDGEDICT_STACK_DW_AT_END) dw_seg = 1 << 20;
    int tail_split = 0;   // measured: 4 -> +0.2..0.7 ms per step (BPTT period 23.6 -> 24.8 us outweighs the shorter tail)
    if (const char* e = getenv("EDGEDICT_STACK_TAIL_SPLIT")) tail_split = atoi(e);

    int margin = margin_launches(d, g, true);
    std::vector<int> next_t(L, 0);          // BPTT steps done; the next one is frame T - 1 - next_t
    std::vector<std::vector<int>> ready_w(L);
    for (int l = 0; l < L; ++l) ready_w[l].assign(g[l].nchunks, l == L - 1 ? 0 : 0x3fffffff);
    // side work of a chunk that layer l's BPTT has just passed (enqueued right after the launch that carries the
    // chunk's last step, launch index w): dX product + LayerNorm backward for the layer below, weight gradients.
    // after_recurrence orders `waiter` behind layer l's recurrence so far: an event on the recurrence stream (ONE
    // shared record per launch, Streams::share - every record costs the next launch ~3.5 us; polling a per-layer
    // done counter from the side streams instead was measured slower: 21.87 vs 21.70 ms per step at E6D2)
    auto after_recurrence = [&](int l, hipStream_t waiter) -> int { return st.chain(st.RS(l), waiter); };
    auto chunk_done = [&](int l, int k, int w) -> int {
            const edgedict_stack_layer_t& y = d->layers[l];
            if (l > 0) {
                // dX_l[chunk] (+)= dG_l[chunk] W_ih, then LayerNorm backward into layer l-1
                const edgedict_stack_layer_t& z = d->layers[l - 1];
                const int t0 = k * g[l].cf, t1 = min(y.T, t0 + g[l].cf);
                const long long r0 = (long long)t0 * B;
                hipStream_t S = st.S[l];
                ED_TRY(after_recurrence(l, S));
                ED_DEV(edgedict_gemm(ED_BF16, ED_BF16, bptr(y.G) + r0 * 4 * H, 4ll * H, 1,
                                     y.wih_t ? y.wih_t : y.wih_p, y.wih_t ? 4ll * H : y.I,
                                     y.wih_t ? 1 : 0, bptr(y.dX) + r0 * y.I, y.I, (t1 - t0) * B, y.I,
                                     4 * H, nullptr, nullptr, y.dX == y.dZ ? 1 : 0, 1, S));
                const int u0 = k * g[l - 1].cf, u1 = min(z.T, u0 + g[l - 1].cf);
                ED_DEV(ed_stack_ln_bwd(bptr(y.dX), (long long)B * y.I, y.I, bptr(z.Yx) + BH,
                                       z.residual ? bptr(z.X) : nullptr, z.ln_gamma, z.mean, z.rstd,
                                       bptr(z.dZ), (float*)(ws + wl.lnpart[l - 1]) + (size_t)k * LNB_GRID * 2 * H,
                                       LNB_GRID, B, H, u0, u1, z.reduce, S, drop_of(z).thresh, drop_of(z).seed,
                                       drop_of(z).scale, drop_of(z).T_out));
                if (soft) ED_DEV(ed_stack_set_flag(bflag + (l - 1) * 512 + k, S));
                else ED_TRY(st.record(Eb[l - 1][k], S));
                queued[l - 1][k] = 1;
                ready_w[l - 1][k] = w + margin;
                if (g_trace) g_trace->chunk_enqueued[g_trace->coff[l - 1] + k] = g_trace->launches;
            }
            // chunks complete from the last to the first: a segment [k, k + dw_seg) is complete when
            // its lowest chunk is (k a multiple of dw_seg, counted so that the LAST segment issued,
            // the one ending at chunk 0, is a full one)
            // Experiment (EDGEDICT_STACK_TAIL_SPLIT=n, off by default): split the two layers that
            // finish LAST once more - frames [nchunks/n * cf, T) issued when the BPTT passes that
            // chunk - so that less work is left for the 0.7 ms tail after the last launch.  Measured
            // slower: the extra products disturb more BPTT launches than the tail shrinks.
            const int k_split = (l <= 1 && tail_split > 1 && g[l].nchunks >= 2 * tail_split && dw_seg >= g[l].nchunks &&
                                 !(d->flags & EDGEDICT_STACK_DW_AT_END))
                                    ? g[l].nchunks / tail_split : 0;
            if (k_split > 0 && k == k_split) {
                ED_TRY(after_recurrence(l, st.W));
                ED_TRY(weight_grads(l, k * g[l].cf, y.T, true, false));
            } else if (k_split > 0 && k == 0) {
                ED_TRY(after_recurrence(l, st.W));
                ED_TRY(weight_grads(l, 0, k_split * g[l].cf, false, l == 0));
            } else if (k_split == 0 && k % dw_seg == 0) {
                if (d->flags & EDGEDICT_STACK_DW_AT_END) {
                    if (k == 0) deferred.push_back(l);
                } else {
                    const int k1 = min(g[l].nchunks, k + dw_seg);
                    const int t0 = k * g[l].cf, t1 = min(y.T, k1 * g[l].cf);
                    ED_TRY(after_recurrence(l, st.W));
                    ED_TRY(weight_grads(l, t0, t1, k1 == g[l].nchunks, l == 0 && k == 0));
                }
            }
        return ED_OK;
    };
    int launches = 0, idle = 0;
    if (st.rt) st.rt->stamp_used[1] = 0;
    if (st.rt && st.rt->tev[1][0]) ED_CHECK_HIP(hipEventRecord(st.rt->tev[1][0], st.R));
    struct Done { int l, k, t; };
    if (sk_ns) {
        // ---- macro-steps: launch w carries, for every runnable layer, its next <= sk_ns BPTT steps (descending t,
        // never across a chunk boundary); the schedule is the one below in units of macro-steps
        const int WGS = (H >> 6) * 4;
        static const int n_cu = [] {
            int dev = 0, n = 256;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
            return n;
        }();
        const int max_slots = max(1, min(ED_STACK_MAX_SLOTS, (g_trace ? 256 : n_cu) / WGS));
        const char* e_m = getenv("EDGEDICT_LPW_MARGIN_B");
        margin = (e_m && atoi(e_m) > 0) ? atoi(e_m) : 2;
        for (int w = 0;; ++w) {
            bool finished = true;
            for (int l = 0; l < L; ++l) finished = finished && next_t[l] >= g[l].T;
            if (finished) break;
            EdSkLaunch Ls;
            Ls.nslot = 0;
            Ls.B = B;
            Ls.H = H;
            Ls.trace = g_wsr_trace;
            Done done[ED_STACK_MAX_SLOTS];
            int ndone = 0;
            // the launch has room for max_slots layers: the layers with the most steps left go first (the full-rate
            // layers at the bottom are the critical path; top-down order let the four layers behind the time
            // reduction take every slot and the two full-rate ones run alone, on half the chip, at the end)
            int order[ED_STACK_MAX_SLOTS];
            for (int l = 0; l < L; ++l) order[l] = l;
            std::stable_sort(order, order + L, [&](int a, int b) { return g[a].T - next_t[a] > g[b].T - next_t[b]; });
            for (int oi = 0; oi < L; ++oi) {
                const int l = order[oi];
                const edgedict_stack_layer_t& y = d->layers[l];
                if (next_t[l] >= g[l].T || Ls.nslot >= max_slots) continue;
                const int t = g[l].T - 1 - next_t[l];
                const int k = t / g[l].cf;
                const bool opens = (t == min(g[l].T, (k + 1) * g[l].cf) - 1);
                if (opens && !(queued[l][k] && w >= ready_w[l][k])) continue;
                int m_min = g[l].m;   // paced only while a faster layer below is in flight
                for (int j = 0; j < l; ++j)
                    if (next_t[j] > 0 && next_t[j] < g[j].T) m_min = min(m_min, g[j].m);
                if (!Pace::allows(w, l, g[l].m, m_min)) continue;
                if (opens && !soft) ED_TRY(st.wait(st.R, Eb[l][k]));
                int t_end = max(k * g[l].cf, t - sk_ns + 1);            // last (lowest) frame of this macro-step
                // a single left-over frame (T = 401 = 25 x 16 + 1 at E6D2) rides along: as a launch of its own it cost the
                // layer - and layer 0 below it, the critical path - a whole launch of the wavefront (37 -> 36 launches)
                if (t_end - k * g[l].cf == 1) t_end = k * g[l].cf;
                EdSkSlot& ss = Ls.slot[Ls.nslot++];
                ss.G = bptr(y.G) + (long long)t * B * 4 * H;
                ss.img = bptr(ws + wl.gimg[l]);
                ss.img_stride = (long long)wl.gimg_stride;
                ss.img_bytes = (long long)(y.T + 1) * (long long)wl.gimg_stride;
                ss.dY = bptr(y.dZ) + (long long)t * BH;
                ss.Cx = y.Cx;
                ss.dC = (float*)(ws + wl.dC[l]);
                ss.Wsk = bptr(y.whh_s);
                ss.part = (float*)(ws + wl.skpart[l]);
                ss.counter = cntb + (size_t)l * LPW_CNT_STRIDE;
                ss.gcounter = gcnt + (size_t)l * SK_CNT_LINES * 64;
                ss.done = (unsigned)next_t[l];
                ss.wait_flag = (soft && opens) ? bflag + l * 512 + k : nullptr;
                ss.t0 = t;
                ss.nsteps = t - t_end + 1;
                ss.T = y.T;
                ss.layer = l;
                if (t_end == k * g[l].cf) {
                    done[ndone].l = l; done[ndone].k = k; done[ndone].t = t_end;
                    ++ndone;
                }
                next_t[l] += t - t_end + 1;
                if (g_trace)
                    for (int tt = t_end; tt <= t; ++tt) g_trace->step_launch[g_trace->toff[l] + tt] = g_trace->launches;
            }
            if (Ls.nslot == 0) {
                ED_CHECK_ARG(++idle < 4096, "encoder_stack: backward schedule made no progress");
                continue;
            }
            idle = 0;
            ++launches;
            Ls.stamp = st.rt ? st.rt->stamp_slot(1, st.R) : nullptr;
            Ls.err = gerr;
            ED_DEV(ed_stack_launch_bwd_sk(Ls, st.R));
            if (g_trace) {
                g_trace->max_slots = max(g_trace->max_slots, Ls.nslot);
                ++g_trace->launches;
            }
            if (ndone > 1) ED_TRY(st.share(st.R));
            for (int i = 0; i < ndone; ++i) ED_TRY(chunk_done(done[i].l, done[i].k, w));
            st.unshare();
        }
    } else
    for (int w = 0;; ++w) {
        bool finished = true;
        for (int l = 0; l < L; ++l) finished = finished && next_t[l] >= g[l].T;
        if (finished) break;
        EdBwdLaunch Lcs[2];
        for (auto& x : Lcs) {
            x.nstep = 0;
            x.B = B; x.H = H;
        }
        Done done[ED_STACK_MAX_SLOTS];
        int ndone = 0;
        for (int l = L - 1; l >= 0; --l) {
            const edgedict_stack_layer_t& y = d->layers[l];
            if (next_t[l] >= g[l].T) continue;
            const int t = g[l].T - 1 - next_t[l];
            const int k = t / g[l].cf;
            const bool opens = (t == min(g[l].T, (k + 1) * g[l].cf) - 1);
            if (opens && !(queued[l][k] && w >= ready_w[l][k])) continue;
            int m_min = g[l].m;   // paced only while a faster layer below is in flight
            for (int j = 0; j < l; ++j)
                if (next_t[j] > 0 && next_t[j] < g[j].T) m_min = min(m_min, g[j].m);
            if (!Pace::allows(w, l, g[l].m, m_min)) continue;
            if (opens && !soft) ED_TRY(st.wait(st.RS(l), Eb[l][k]));
            EdBwdLaunch& Lc = Lcs[st.which(l)];
            EdBwdStep& sl = Lc.step[Lc.nstep++];
            sl.wait_flag = (soft && opens) ? bflag + l * 512 + k : nullptr;
            bf16_t* f0 = bptr(ws + wl.frag0[l]);
            bf16_t* f1 = bptr(ws + wl.frag1[l]);
            sl.G_t = bptr(y.G) + (long long)t * B * 4 * H;
            sl.gfrag_in = (t == g[l].T - 1) ? nullptr : (((t + 1) & 1) ? f1 : f0);
            sl.gfrag_out = (t == 0) ? nullptr : ((t & 1) ? f1 : f0);
            sl.dY_t = bptr(y.dZ) + (long long)t * BH;
            sl.C_t = y.Cx + (long long)(t + 1) * BH;
            sl.C_prev = y.Cx + (long long)t * BH;
            sl.dC = (float*)(ws + wl.dC[l]);
            sl.WTfrag = bptr(y.whh_b);
            if (t == k * g[l].cf) {
                done[ndone].l = l; done[ndone].k = k; done[ndone].t = t;
                ++ndone;
            }
            ++next_t[l];
            if (g_trace) g_trace->step_launch[g_trace->toff[l] + t] = g_trace->launches;
        }
        if (Lcs[0].nstep + Lcs[1].nstep == 0) {
            ED_CHECK_ARG(++idle < 4096, "encoder_stack: backward schedule made no progress");
            continue;
        }
        idle = 0;
        ++launches;
        Lcs[0].stamp = st.rt ? st.rt->stamp_slot(1, st.R) : nullptr;
        Lcs[1].stamp = nullptr;
        Lcs[0].err = Lcs[1].err = gerr;
        ED_DEV(ed_stack_launch_bwd(Lcs[0], st.R));
        if (st.R2 != st.R) ED_DEV(ed_stack_launch_bwd(Lcs[1], st.R2));
        if (g_trace) {
            g_trace->max_slots = max(g_trace->max_slots, Lcs[0].nstep + Lcs[1].nstep);
            ++g_trace->launches;
        }
        for (int i = 0; i < ndone; ++i) ED_TRY(chunk_done(done[i].l, done[i].k, w));
    }
    const int Wtot = launches;
    if (st.rt && st.rt->tev[1][1]) {
        ED_CHECK_HIP(hipEventRecord(st.rt->tev[1][1], st.R));
        st.rt->tlaunches[1] = Wtot;
    }
    // ---- input LayerNorm parameters: dX_0 = dG_0 W_ih (all frames), then the two column sums
    {
        const edgedict_stack_layer_t& y = d->layers[0];
        bf16_t* dX0 = bptr(ws + wl.dX0);
        ED_TRY(st.chain(st.RS(0), st.S[0]));
        ED_DEV(edgedict_gemm(ED_BF16, ED_BF16, y.G, 4ll * H, 1, y.wih_t ? y.wih_t : y.wih_p,
                             y.wih_t ? 4ll * H : y.I, y.wih_t ? 1 : 0, dX0, y.I, y.T * B, y.I, 4 * H,
                             nullptr, nullptr, 0, 1, st.S[0]));
        ED_DEV(ed_stack_input_norm_bwd(d->x_dtype, d->x, dX0, d->in_mean, d->in_rstd, d->d_in_gamma,
                                       d->d_in_beta, B, d->T0, d->I0, st.S[0]));
        // LayerNorm parameter gradients: sum the per-workgroup partial rows of every launch
        for (int l = 1; l < L; ++l) ED_TRY(st.chain(st.S[l], st.S[0]));
        for (int l = 0; l < L; ++l) {
            const edgedict_stack_layer_t& z = d->layers[l];
            const float* part = (const float*)(ws + wl.lnpart[l]);
            ED_DEV(ed_stack_sum_parts(part, g[l].nchunks * LNB_GRID, H, z.dgamma, z.dbeta, st.S[0]));
        }
    }
    if (!deferred.empty()) {
        ED_TRY(st.chain(st.R, st.W));
        if (st.R2 != st.R) ED_TRY(st.chain(st.R2, st.W));
        for (int l : deferred) ED_TRY(weight_grads(l, 0, d->layers[l].T, true, false));
    }
    ED_TRY(st.chain(st.R, st.C));
    if (st.R2 != st.R) ED_TRY(st.chain(st.R2, st.C));
    for (int l = 0; l < L; ++l) ED_TRY(st.chain(st.S[l], st.C));
    ED_TRY(st.chain(st.W, st.C));
    return ED_OK;
}
