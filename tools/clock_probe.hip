// Shader clock under load: s_memtime (shader cycles) against s_memrealtime (100 MHz) around a loop of MFMAs on
// random / zero operands, or of plain VALU work, on every CU.   hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__global__ __launch_bounds__(512, 1) void probe(int mode, int iters, unsigned seed, long long* out, float* sink) {
    union { bf16x8_t v; unsigned u[4]; } a[4], b[4];
    unsigned s = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x;
    for (int i = 0; i < 4; ++i)
        for (int q = 0; q < 4; ++q) {
            s = s * 1664525u + 1013904223u;
            // bf16 pairs in [-1, 1): sign + exponent 0x3f.. region
            unsigned hi = 0x3f000000u | (s & 0x807f0000u), lo = 0x3f00u | ((s >> 3) & 0x807fu);
            a[i].u[q] = mode == 1 ? 0u : (hi | lo);
            s = s * 1664525u + 1013904223u;
            hi = 0x3f000000u | (s & 0x807f0000u); lo = 0x3f00u | ((s >> 3) & 0x807fu);
            b[i].u[q] = mode == 1 ? 0u : (hi | lo);
        }
    f32x4_t acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float x = threadIdx.x;
    __syncthreads();
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (mode <= 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].v, b[j].v, acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 64; ++i) x = x * 1.0001f + 0.5f;
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float t = x;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][3];
    if (t == 12345.678f) sink[0] = t;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = w1 - w0; }
}

int main() {
    long long* d; float* sink;
    hipMalloc(&d, 256 * 2 * sizeof(long long)); hipMalloc(&sink, 4);
    const char* names[3] = {"MFMA random", "MFMA zeros ", "VALU only  "};
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
        const int iters = 200000;      // 16 MFMA x 16 cycles x 2 waves per SIMD = 512 cycles per iteration
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, mode, iters, 7u, d, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(512);
        hipMemcpy(h.data(), d, 512 * sizeof(long long), hipMemcpyDeviceToHost);
        double cs = 0, ws = 0;
        for (int i = 0; i < 256; ++i) { cs += h[2 * i]; ws += h[2 * i + 1]; }
        const double mhz = cs / ws * 100.0;
        const double tf = mode <= 1 ? 256.0 * 8 * iters * 16 * 16384.0 / (ms * 1e-3) / 1e12 : 0.0;
        printf("%s  %8.2f ms   clock64/wall_clock64 -> %7.1f MHz (if s_memtime counts shader cycles)   %7.1f TFLOP/s\n",
               names[mode], ms, mhz, tf);
    }
    return 0;
}
