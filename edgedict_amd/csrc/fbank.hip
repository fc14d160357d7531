// Fused log-mel filterbank front-end for gfx950.
//
// Replaces FilterbankFeatures.forward (rnnt/features.py:106-152; twin parts/features.py:298-347)
// followed by Downsample.forward (rnnt/transforms.py:38-51):
//   dither (separate in-place kernel) -> pre-emphasis y[n] = x[n] - a*x[n-1] -> torch.stft with
//   center=True / reflect padding, hann window of win_length centred in n_fft -> |.|^2 ->
//   mel filterbank matmul -> log(x + 1e-20) -> zero frames t >= ceil(N/hop) -> stack `stack`
//   consecutive frames into one feature vector (zero frames appended to a multiple of `stack`).
// The spectrum never reaches HBM: one wave64 owns one frame; windowed samples go straight into
// LDS in bit-reversed order, a radix-2 FFT runs in place in LDS, the power spectrum is reduced
// against the (sparse, triangular) mel rows and the 80 log-energies are written directly in the
// layout the encoder wants.  The kernel is launch/HBM-bound (reads ~1 MB, writes ~0.4 MB per 15 s
// utterance); no MFMA on purpose.
#include "common.hpp"

namespace {

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// x[b, n] += amp * N(0,1), counter-based (seed, b, n) so results do not depend on the launch shape
__global__ void dither_kernel(float* __restrict__ x, long long stride, int B, int N,
                              const int32_t* __restrict__ lengths, float amp, unsigned seed) {
    const long long total = (long long)B * N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / N), n = (int)(i % N);
        if (lengths && n >= lengths[b]) continue;
        const unsigned h1 = hash32(seed ^ hash32((unsigned)b * 0x9e3779b9U + (unsigned)n));
        const unsigned h2 = hash32(h1 + 0x85ebca6bU);
        const float u1 = ((h1 >> 8) + 1) * (1.0f / 16777217.0f);  // (0,1]
        const float u2 = (h2 >> 8) * (1.0f / 16777216.0f);
        const float g = sqrtf(-2.f * logf(u1)) * cospif(2.f * u2);
        x[(long long)b * stride + n] += amp * g;
    }
}

struct FbankArgs {
    const float* wave;      // [B, >=N] row stride wave_stride
    const int32_t* lengths; // nullable [B]: valid samples per utterance
    const float* window;    // [n_fft] window already centred/zero-padded
    const float* twiddle;   // [n_fft/2][2] cos, sin of 2*pi*k/n_fft
    const float* fb;        // [n_mels, n_fft/2+1]
    const int32_t* fb_range;// [n_mels][2] first / one-past-last non-zero bin
    void* out;
    long long wave_stride;
    long long o_b, o_group, o_k, o_m;  // out[b*o_b + (f/stack)*o_group + (f%stack)*o_k + m*o_m]
    int B, N, n_fft, log2n, hop, n_mels, stack, frames_out;  // frames_out: multiple of stack
    int win_lo, win_hi;     // non-zero window support [lo, hi)
    float preemph;
    int do_log;
    int mask_only;          // 1: `lengths` only masks frames >= ceil(len/hop); the signal is the whole
                            //    padded row (parts/features.py:298-336 semantics); 0: the row ENDS at
                            //    lengths[b] (per-utterance transform, rnnt/dataset.py:102-103)
    int n_signal;           // samples [n_signal, N) are zero padding appended AFTER the pre-emphasis
                            // (parts/features.py:289-294 pads the pre-emphasised short signal)
    int copies;             // the feature vector is written `copies` times, o_copy apart
    long long o_copy;       // (parts/features.py:111-123 frame "splicing")
};

template <typename TO>
__global__ __launch_bounds__(256) void fbank_kernel(FbankArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // per wave: re[n_fft], im[n_fft]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_fft = a.n_fft, half = n_fft >> 1, nbins = half + 1;
    float* re = lds + (size_t)wave * 2 * n_fft;
    float* im = re + n_fft;
    const long long frame_id = (long long)blockIdx.x * 4 + wave;
    const long long total = (long long)a.B * a.frames_out;
    const bool active = frame_id < total;
    const int b = active ? (int)(frame_id / a.frames_out) : 0;
    const int f = active ? (int)(frame_id % a.frames_out) : 0;
    const int Lb = a.lengths ? min(a.lengths[b], a.N) : a.N;
    const int Nb = a.mask_only ? a.N : Lb;
    const int n_frames = Nb > 0 ? 1 + Nb / a.hop : 0;  // torch.stft(center=True)
    const int seq_len = (Lb + a.hop - 1) / a.hop;      // get_seq_len: ceil(N / hop)
    const bool compute = active && f < n_frames && f < seq_len;

    if (compute) {
        const float* x = a.wave + (long long)b * a.wave_stride;
        const int start = f * a.hop - half;  // centre padding of n_fft/2
        for (int i = lane; i < n_fft; i += 64) {
            float v = 0.f;
            if (i >= a.win_lo && i < a.win_hi) {
                int n = start + i;
                if (n < 0) n = -n;                       // reflect (no edge repeat)
                if (n >= Nb) n = 2 * (Nb - 1) - n;
                n = min(max(n, 0), Nb - 1);
                const float cur = x[n];
                const float y = (n > 0) ? cur - a.preemph * x[n - 1] : cur;
                v = (n < a.n_signal) ? y * a.window[i] : 0.f;
            }
            const int r = (int)(__brev((unsigned)i) >> (32 - a.log2n));
            re[r] = v;
            im[r] = 0.f;
        }
    }
    __syncthreads();
    // in-place radix-2 DIT, n_fft/2 butterflies per stage shared by the wave's 64 lanes
    for (int s = 0; s < a.log2n; ++s) {
        const int h = 1 << s;
        if (compute) {
            for (int j = lane; j < half; j += 64) {
                const int pos = j & (h - 1);
                const int i0 = ((j >> s) << (s + 1)) + pos;
                const int i1 = i0 + h;
                const int tw = pos << (a.log2n - 1 - s);
                const float c = a.twiddle[2 * tw], sn = -a.twiddle[2 * tw + 1];  // exp(-i*theta)
                const float br = re[i1] * c - im[i1] * sn;
                const float bi = re[i1] * sn + im[i1] * c;
                const float ar = re[i0], ai = im[i0];
                re[i0] = ar + br; im[i0] = ai + bi;
                re[i1] = ar - br; im[i1] = ai - bi;
            }
        }
        __syncthreads();
    }
    if (compute) {
        for (int k = lane; k < nbins; k += 64) re[k] = re[k] * re[k] + im[k] * im[k];
    }
    __syncthreads();
    if (!active) return;
    TO* out = reinterpret_cast<TO*>(a.out);
    const long long obase = (long long)b * a.o_b + (long long)(f / a.stack) * a.o_group +
                            (long long)(f % a.stack) * a.o_k;
    for (int m = lane; m < a.n_mels; m += 64) {
        float v = 0.f;
        if (compute) {
            const int lo = a.fb_range[2 * m], hi = a.fb_range[2 * m + 1];
            const float* w = a.fb + (long long)m * nbins;
            float acc = 0.f;
            for (int k = lo; k < hi; ++k) acc += w[k] * re[k];
            v = a.do_log ? logf(acc + 1e-20f) : acc;
        }
        for (int c = 0; c < a.copies; ++c)
            ElemIO<TO>::store(out + obase + c * a.o_copy + (long long)m * a.o_m, v);
    }
}

// normalize_batch of parts/features.py:80-109 (rnnt/features.py:7-30) on x[b, row, f] (f32,
// frame stride 1): mean and UNBIASED std over the first n_b = min(ceil(len_b / hop), F) frames -
// per feature row (mode 1: one workgroup per (b, row)) or over all rows of the utterance (mode 2:
// one workgroup per b) - then (x - mean) / (std + 1e-5) on those frames (later frames are masked
// to zero by the caller's contract and stay untouched).  Two-pass, fp32, HBM/launch-bound.
__device__ __forceinline__ float block_sum256(float v, float* part) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    return part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(256) void feat_normalize_kernel(
    float* __restrict__ x, const int32_t* __restrict__ lengths, int N, int hop, int rows, int F,
    long long o_b, long long o_row, int mode) {
    __shared__ float part[4];
    const int b = mode == 1 ? blockIdx.x / rows : blockIdx.x;
    const int r0 = mode == 1 ? blockIdx.x % rows : 0;
    const int nr = mode == 1 ? 1 : rows;
    const int Lb = lengths ? min(lengths[b], N) : N;
    const int n = min((Lb + hop - 1) / hop, F);
    if (n <= 0) return;
    float* base = x + (long long)b * o_b + (long long)r0 * o_row;
    const long long cnt = (long long)nr * n;
    float s = 0.f;
    for (long long i = threadIdx.x; i < cnt; i += 256) s += base[(i / n) * o_row + (i % n)];
    const float mean = block_sum256(s, part) / (float)cnt;
    float q = 0.f;
    for (long long i = threadIdx.x; i < cnt; i += 256) {
        const float d = base[(i / n) * o_row + (i % n)] - mean;
        q += d * d;
    }
    const float var = block_sum256(q, part) / (float)(cnt - 1);   // cnt == 1 -> NaN, as torch.std
    const float inv = 1.f / (sqrtf(var) + 1e-5f);
    for (long long i = threadIdx.x; i < cnt; i += 256) {
        float* e = base + (i / n) * o_row + (i % n);
        *e = (*e - mean) * inv;
    }
}

}  // namespace

extern "C" int edgedict_dither(float* wave, long long wave_stride, int B, int N,
                               const int32_t* lengths, float amplitude, unsigned seed,
                               void* stream_) {
    ED_CHECK_ARG(B >= 0 && N >= 0, "dither: bad shape");
    if (B == 0 || N == 0 || amplitude == 0.f) return ED_OK;
    ED_CHECK_ARG(wave, "dither: null pointer");
    hipLaunchKernelGGL(dither_kernel, dim3(ed_grid_for((long long)B * N, 256 * 4)), dim3(256), 0,
                       (hipStream_t)stream_, wave, wave_stride, B, N, lengths, amplitude, seed);
    ED_CHECK_LAUNCH("dither");
    return ED_OK;
}

static int fbank_launch(const float* wave, long long wave_stride, int B, int N,
                                      const int32_t* lengths, const float* window,
                                      const float* twiddle, const float* fb,
                                      const int32_t* fb_range, int n_fft, int win_lo, int win_hi,
                                      int hop, int n_mels, float preemph, int do_log, void* out,
                                      int out_dtype, long long o_b, long long o_group,
                                      long long o_k, long long o_m, int stack, int frames_out,
                                      int mask_only, int copies, long long o_copy, int n_signal,
                                      void* stream_) {
    ED_CHECK_ARG(out_dtype == ED_F32 || out_dtype == ED_BF16, "fbank: bad output dtype");
    ED_CHECK_ARG(B >= 0 && N >= 0 && hop > 0 && n_mels > 0 && stack > 0 && copies >= 1, "fbank: bad shape");
    int log2n = 0;
    while ((1 << log2n) < n_fft) ++log2n;
    ED_CHECK_ARG((1 << log2n) == n_fft && n_fft >= 64 && n_fft <= 2048,
                 "fbank: n_fft = %d must be a power of two in [64, 2048]", n_fft);
    ED_CHECK_ARG(frames_out >= 0 && frames_out % stack == 0, "fbank: frames_out must be a multiple of stack");
    ED_CHECK_ARG(win_lo >= 0 && win_lo <= win_hi && win_hi <= n_fft, "fbank: bad window support");
    if (B == 0 || frames_out == 0) return ED_OK;
    ED_CHECK_ARG(wave && window && twiddle && fb && fb_range && out, "fbank: null pointer");
    FbankArgs a;
    a.wave = wave; a.lengths = lengths; a.window = window; a.twiddle = twiddle; a.fb = fb;
    a.fb_range = fb_range; a.out = out; a.wave_stride = wave_stride;
    a.o_b = o_b; a.o_group = o_group; a.o_k = o_k; a.o_m = o_m;
    a.B = B; a.N = N; a.n_fft = n_fft; a.log2n = log2n; a.hop = hop; a.n_mels = n_mels;
    a.stack = stack; a.frames_out = frames_out; a.win_lo = win_lo; a.win_hi = win_hi;
    a.preemph = preemph; a.do_log = do_log;
    a.mask_only = mask_only; a.copies = copies; a.o_copy = o_copy; a.n_signal = n_signal;
    const long long total = (long long)B * frames_out;
    const long long blocks = (total + 3) / 4;
    ED_CHECK_ARG(blocks < (1ll << 31), "fbank: too many frames");
    const size_t lds = (size_t)4 * 2 * n_fft * sizeof(float);
    hipStream_t s = (hipStream_t)stream_;
    if (out_dtype == ED_F32)
        hipLaunchKernelGGL(fbank_kernel<float>, dim3((unsigned)blocks), dim3(256), lds, s, a);
    else
        hipLaunchKernelGGL(fbank_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), lds, s, a);
    ED_CHECK_LAUNCH("fbank");
    return ED_OK;
}

extern "C" int edgedict_fbank_forward(const float* wave, long long wave_stride, int B, int N,
                                      const int32_t* lengths, const float* window,
                                      const float* twiddle, const float* fb,
                                      const int32_t* fb_range, int n_fft, int win_lo, int win_hi,
                                      int hop, int n_mels, float preemph, int do_log, void* out,
                                      int out_dtype, long long o_b, long long o_group,
                                      long long o_k, long long o_m, int stack, int frames_out,
                                      void* stream_) {
    return fbank_launch(wave, wave_stride, B, N, lengths, window, twiddle, fb, fb_range, n_fft, win_lo,
                        win_hi, hop, n_mels, preemph, do_log, out, out_dtype, o_b, o_group, o_k, o_m,
                        stack, frames_out, 0, 1, 0, N, stream_);
}

extern "C" int edgedict_fbank_forward_masked(const float* wave, long long wave_stride, int B, int N,
                                             const int32_t* seq_len, const float* window,
                                             const float* twiddle, const float* fb,
                                             const int32_t* fb_range, int n_fft, int win_lo,
                                             int win_hi, int hop, int n_mels, float preemph,
                                             int do_log, float* out, long long o_b, long long o_m,
                                             int frames_out, int copies, long long o_copy,
                                             int normalize, int n_signal, void* stream_) {
    ED_CHECK_ARG(normalize >= 0 && normalize <= 2, "fbank_masked: normalize must be 0 (none), 1 (per_feature) or 2 (all_features)");
    const int rc = fbank_launch(wave, wave_stride, B, N, seq_len, window, twiddle, fb, fb_range, n_fft,
                                win_lo, win_hi, hop, n_mels, preemph, do_log, out, ED_F32, o_b, 1, 0,
                                o_m, 1, frames_out, 1, copies, o_copy,
                                n_signal > 0 && n_signal < N ? n_signal : N, stream_);
    if (rc != ED_OK || normalize == 0 || B == 0 || frames_out == 0) return rc;
    const int rows = n_mels * copies;
    ED_CHECK_ARG(o_copy == (long long)n_mels * o_m || copies == 1, "fbank_masked: copies must be adjacent row blocks");
    const int F = min(frames_out, 1 + N / hop);
    hipLaunchKernelGGL(feat_normalize_kernel, dim3(normalize == 1 ? (unsigned)(B * rows) : (unsigned)B),
                       dim3(256), 0, (hipStream_t)stream_, out, seq_len, N, hop, rows, F, o_b, o_m, normalize);
    ED_CHECK_LAUNCH("fbank_masked normalize");
    return ED_OK;
}
