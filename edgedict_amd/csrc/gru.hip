// GRU time recurrence (forward and BPTT) for gfx950: the encoder variant the reference selects
// with module_type='GRU' (ResLayerNormGRU, rnnt/models.py:77-116: one 1-layer batch_first nn.GRU per
// encoder layer).  PyTorch cell semantics, gate order r,z,n in the 3H rows of weight_ih / weight_hh:
//   r = sigmoid(W_ir x + b_ir + W_hr h + b_hr)      z = sigmoid(W_iz x + b_iz + W_hz h + b_hz)
//   n = tanh(W_in x + b_in + r * (W_hn h + b_hn))   h' = (1 - z) * n + z * h
// Same division of labour as lstm.hip: the input product x W_ih^T + b_ih for all timesteps is one
// MFMA GEMM that leaves G[B,T,3H]; the serial part h_{t-1} W_hh^T is ONE SMALL KERNEL PER TIMESTEP
// (operands straight from L2 into MFMA registers, K split over the 4 waves, LDS reduction, cell math
// per (row, unit)).  This is the generic per-layer path only (fp32 parity mode and bf16): the GRU
// variant is a compatibility module, not the north-star configuration, and has no fragment-image
// fast path or layer wavefront.
// Backward step t: dh = dY[:,t] + z_{t+1} dh_{t+1} (carried) + DH[:,t+1] W_hh (K = 3H), then
//   dn = dh (1-z), dz = dh (h_prev - n); pre-activation gradients
//   a_n = dn (1-n^2), a_z = dz z (1-z), a_r = a_n hn r (1-r)     (hn = W_hn h_prev + b_hn, saved)
//   input side  G[:,t]  <- [a_r, a_z, a_n]        (-> dX, dW_ih, db_ih by GEMM / column sum)
//   hidden side DH[:,t] <- [a_r, a_z, a_n r]      (-> dW_hh, db_hh; operand of the next step)
#include "common.hpp"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ float gsig(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ bf16x8_t gfrag(const bf16_t* row_ptr, int k, int kmax, bool row_ok) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row_ok && k + 8 <= kmax) v = *reinterpret_cast<const uint4*>(row_ptr + k);
    return *reinterpret_cast<bf16x8_t*>(&v);
}
__device__ __forceinline__ float gfrag(const float* row_ptr, int k, int kmax, bool row_ok) {
    return (row_ok && k < kmax) ? row_ptr[k] : 0.f;
}
__device__ __forceinline__ f32x4_t gmma(bf16x8_t a, bf16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t gmma(float a, float b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
template <typename T> struct GK;
template <> struct GK<bf16_t> { static constexpr int K = 32, LANE_K = 8; };
template <> struct GK<float> { static constexpr int K = 4, LANE_K = 1; };

constexpr int GU = 4;   // hidden units per forward workgroup: 3 gates x 4 units = 12 of a 16-wide N tile

template <typename T>
__global__ __launch_bounds__(256) void gru_step_fwd(
    T* __restrict__ G, T* __restrict__ Hprev, T* __restrict__ Y, T* __restrict__ HN,
    const T* __restrict__ Whh, const float* __restrict__ bhh, float* __restrict__ hN, int B, int Tn,
    int H, int t) {
    __shared__ float red[4][64][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j0 = blockIdx.x * GU;
    const int b0 = blockIdx.y * 64;
    constexpr int MKK = GK<T>::K, LK = GK<T>::LANE_K;
    const int steps_total = (H + MKK - 1) / MKK;
    const int steps_per_wave = (steps_total + 3) / 4;
    const int kbeg = wave * steps_per_wave * MKK;
    const int kend = min(H, kbeg + steps_per_wave * MKK);
    // B operand rows: n = gate*GU + unit (n < 12) -> W_hh row gate*H + j0 + unit
    const int n = lane & 15;
    const bool w_ok = n < 3 * GU && (j0 + (n % GU)) < H;
    const int wrow = w_ok ? (n / GU) * H + j0 + (n % GU) : 0;
    const T* wptr = Whh + (long long)wrow * H;
    const int koff = (lane >> 4) * LK;
    f32x4_t acc[4];
    const T* aptr[4];
    bool a_ok[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        acc[m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const int b = b0 + m * 16 + (lane & 15);
        a_ok[m] = b < B;
        aptr[m] = Hprev + ((long long)min(b, B - 1) * Tn + t) * H;
    }
    for (int k = kbeg; k < kend; k += MKK) {
        const auto bf = gfrag(wptr, k + koff, H, w_ok);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = gmma(gfrag(aptr[m], k + koff, H, a_ok[m]), bf, acc[m]);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][m * 16 + (lane >> 4) * 4 + r][lane & 15] = acc[m][r];
    __syncthreads();

    const int bl = threadIdx.x / GU, u = threadIdx.x % GU;
    const int b = b0 + bl, j = j0 + u;
    if (b >= B || j >= H) return;
    const long long row = (long long)b * Tn + t;
    float hh[3];
#pragma unroll
    for (int gate = 0; gate < 3; ++gate) {
        const int col = gate * GU + u;
        hh[gate] = red[0][bl][col] + red[1][bl][col] + red[2][bl][col] + red[3][bl][col] + bhh[gate * H + j];
    }
    T* grow = G + row * 3 * H;
    const float r = gsig(ElemIO<T>::load(grow + j) + hh[0]);
    const float z = gsig(ElemIO<T>::load(grow + H + j) + hh[1]);
    const float nn = tanhf(ElemIO<T>::load(grow + 2 * H + j) + r * hh[2]);
    const float hp = ElemIO<T>::load(Hprev + row * H + j);
    const float h = (1.f - z) * nn + z * hp;
    ElemIO<T>::store(grow + j, r);
    ElemIO<T>::store(grow + H + j, z);
    ElemIO<T>::store(grow + 2 * H + j, nn);
    ElemIO<T>::store(HN + row * H + j, hh[2]);
    ElemIO<T>::store(Y + row * H + j, h);
    if (t + 1 < Tn) ElemIO<T>::store(Hprev + (row + 1) * H + j, h);
    else if (hN) hN[(long long)b * H + j] = h;
}

template <typename T>
__global__ void gru_init_hprev(T* __restrict__ Hprev, const float* __restrict__ h0, int B, int Tn, int H) {
    const long long n = (long long)B * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / H, j = i % H;
        ElemIO<T>::store(Hprev + (b * Tn) * H + j, h0 ? h0[i] : 0.f);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gru_step_bwd(
    T* __restrict__ G, T* __restrict__ DH, const T* __restrict__ dY, const T* __restrict__ Hprev,
    const T* __restrict__ HN, const T* __restrict__ WhhT, float* __restrict__ dhz, int B, int Tn,
    int H, int t) {
    __shared__ float red[4][16][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j0 = blockIdx.x * 16;
    const int b0 = blockIdx.y * 16;
    constexpr int MKK = GK<T>::K, LK = GK<T>::LANE_K;
    const int K = 3 * H;
    f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (t + 1 < Tn) {
        const int steps_total = (K + MKK - 1) / MKK;
        const int steps_per_wave = (steps_total + 3) / 4;
        const int kbeg = wave * steps_per_wave * MKK;
        const int kend = min(K, kbeg + steps_per_wave * MKK);
        const int rb = b0 + (lane & 15);
        const bool a_ok = rb < B;
        const T* aptr = DH + ((long long)min(rb, B - 1) * Tn + t + 1) * K;
        const int jn = j0 + (lane & 15);
        const bool w_ok = jn < H;
        const T* wptr = WhhT + (long long)min(jn, H - 1) * K;
        const int koff = (lane >> 4) * LK;
        for (int k = kbeg; k < kend; k += MKK)
            acc = gmma(gfrag(aptr, k + koff, K, a_ok), gfrag(wptr, k + koff, K, w_ok), acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(lane >> 4) * 4 + r][lane & 15] = acc[r];
    __syncthreads();
    const int bl = threadIdx.x >> 4, nl = threadIdx.x & 15;
    const int b = b0 + bl, j = j0 + nl;
    if (b >= B || j >= H) return;
    const long long row = (long long)b * Tn + t;
    float dh = red[0][bl][nl] + red[1][bl][nl] + red[2][bl][nl] + red[3][bl][nl] + dhz[(long long)b * H + j];
    if (dY) dh += ElemIO<T>::load(dY + row * H + j);
    T* grow = G + row * K;
    const float r = ElemIO<T>::load(grow + j), z = ElemIO<T>::load(grow + H + j),
                nn = ElemIO<T>::load(grow + 2 * H + j);
    const float hn = ElemIO<T>::load(HN + row * H + j), hp = ElemIO<T>::load(Hprev + row * H + j);
    const float a_n = dh * (1.f - z) * (1.f - nn * nn);
    const float a_z = dh * (hp - nn) * z * (1.f - z);
    const float a_r = a_n * hn * r * (1.f - r);
    ElemIO<T>::store(grow + j, a_r);
    ElemIO<T>::store(grow + H + j, a_z);
    ElemIO<T>::store(grow + 2 * H + j, a_n);
    T* drow = DH + row * K;
    ElemIO<T>::store(drow + j, a_r);
    ElemIO<T>::store(drow + H + j, a_z);
    ElemIO<T>::store(drow + 2 * H + j, a_n * r);
    dhz[(long long)b * H + j] = dh * z;
}

template <typename T>
int gru_fwd(void* G, void* Hprev, void* Y, void* HN, const void* Whh, const float* bhh, const float* h0,
            float* hN, int B, int Tn, int H, hipStream_t s) {
    hipLaunchKernelGGL(gru_init_hprev<T>, dim3(ed_grid_for((long long)B * H, 256)), dim3(256), 0, s,
                       (T*)Hprev, h0, B, Tn, H);
    dim3 grid((H + GU - 1) / GU, (B + 63) / 64);
    for (int t = 0; t < Tn; ++t)
        hipLaunchKernelGGL(gru_step_fwd<T>, grid, dim3(256), 0, s, (T*)G, (T*)Hprev, (T*)Y, (T*)HN,
                           (const T*)Whh, bhh, hN, B, Tn, H, t);
    ED_CHECK_LAUNCH("gru_step_fwd");
    return ED_OK;
}

template <typename T>
int gru_bwd(void* G, void* DH, const void* dY, const void* Hprev, const void* HN, const void* WhhT,
            float* dhz, int B, int Tn, int H, hipStream_t s) {
    ED_CHECK_HIP(hipMemsetAsync(dhz, 0, (size_t)B * H * sizeof(float), s));
    dim3 grid((H + 15) / 16, (B + 15) / 16);
    for (int t = Tn - 1; t >= 0; --t)
        hipLaunchKernelGGL(gru_step_bwd<T>, grid, dim3(256), 0, s, (T*)G, (T*)DH, (const T*)dY,
                           (const T*)Hprev, (const T*)HN, (const T*)WhhT, dhz, B, Tn, H, t);
    ED_CHECK_LAUNCH("gru_step_bwd");
    return ED_OK;
}

}  // namespace

extern "C" int edgedict_gru_forward(int dtype, void* G, void* Hprev, void* Y, void* HN, const void* Whh,
                                    const float* b_hh, const float* h0, float* hN, int B, int T, int H,
                                    void* stream) {
    ED_CHECK_ARG(dtype == ED_F32 || dtype == ED_BF16, "gru_forward: bad dtype %d", dtype);
    ED_CHECK_ARG(B > 0 && T > 0 && H > 0, "gru_forward: B,T,H must be positive (got %d,%d,%d)", B, T, H);
    ED_CHECK_ARG(H % 8 == 0, "gru_forward: hidden size %d must be a multiple of 8", H);
    ED_CHECK_ARG(G && Hprev && Y && HN && Whh && b_hh, "gru_forward: null pointer");
    if (dtype == ED_F32)
        return gru_fwd<float>(G, Hprev, Y, HN, Whh, b_hh, h0, hN, B, T, H, (hipStream_t)stream);
    return gru_fwd<bf16_t>(G, Hprev, Y, HN, Whh, b_hh, h0, hN, B, T, H, (hipStream_t)stream);
}

extern "C" int edgedict_gru_backward(int dtype, void* G, void* DH, const void* dY, const void* Hprev,
                                     const void* HN, const void* WhhT, float* dh_ws, int B, int T, int H,
                                     void* stream) {
    ED_CHECK_ARG(dtype == ED_F32 || dtype == ED_BF16, "gru_backward: bad dtype %d", dtype);
    ED_CHECK_ARG(B > 0 && T > 0 && H > 0, "gru_backward: B,T,H must be positive");
    ED_CHECK_ARG(H % 8 == 0, "gru_backward: hidden size %d must be a multiple of 8", H);
    ED_CHECK_ARG(G && DH && Hprev && HN && WhhT && dh_ws, "gru_backward: null pointer");
    if (dtype == ED_F32)
        return gru_bwd<float>(G, DH, dY, Hprev, HN, WhhT, dh_ws, B, T, H, (hipStream_t)stream);
    return gru_bwd<bf16_t>(G, DH, dY, Hprev, HN, WhhT, dh_ws, B, T, H, (hipStream_t)stream);
}
