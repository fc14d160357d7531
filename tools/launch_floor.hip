// Micro-benchmark: cost of a chain of dependent tiny kernels on one stream (the structure of the
// LSTM recurrence).  hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <chrono>

__global__ void empty_k(float* p) {}
__global__ void touch_k(float* p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] + 1.f;
}
// each block reads `bytes_per_block` from a shared region (L2-resident), writes a little
__global__ __launch_bounds__(256) void read_k(const uint4* __restrict__ src, float* dst, int vec_per_thread) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    const uint4* s = src + threadIdx.x;
#pragma unroll 8
    for (int i = 0; i < vec_per_thread; ++i) {
        uint4 v = s[i * 256];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) dst[blockIdx.x] = 1.f;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename F> double time_chain(F launch, int n, hipStream_t s) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 50; ++i) launch(i);
    hipStreamSynchronize(s);
    auto t0 = std::chrono::high_resolution_clock::now();
    hipEventRecord(a, s);
    for (int i = 0; i < n; ++i) launch(i);
    hipEventRecord(b, s);
    auto t1 = std::chrono::high_resolution_clock::now();
    hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, a, b);
    double host_us = std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
    printf("   [host enqueue %.2f us/launch] ", host_us);
    return ms * 1000.0 / n;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    float* p; CK(hipMalloc(&p, 64 << 20));
    CK(hipMemset(p, 0, 64 << 20));
    const int N = 2000;
    double t;
    t = time_chain([&](int) { hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, s, p); }, N, s);
    printf("empty 1 block:            %.2f us/kernel\n", t);
    t = time_chain([&](int) { hipLaunchKernelGGL(empty_k, dim3(256), dim3(256), 0, s, p); }, N, s);
    printf("empty 256x256:            %.2f us/kernel\n", t);
    t = time_chain([&](int) { hipLaunchKernelGGL(touch_k, dim3(256), dim3(256), 0, s, p, 65536); }, N, s);
    printf("touch 256KB rw:           %.2f us/kernel\n", t);
    for (int vpt : {8, 40, 160}) {
        t = time_chain([&](int) { hipLaunchKernelGGL(read_k, dim3(256), dim3(256), 0, s, (const uint4*)p, p + (8 << 20), vpt); }, N, s);
        printf("read %4d KB/block (same region, 256 blocks): %.2f us/kernel\n", vpt * 4, t);
    }
    // graph replay of the same chain
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 500; ++i) hipLaunchKernelGGL(touch_k, dim3(256), dim3(256), 0, s, p, 65536);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, s);
    for (int r = 0; r < 4; ++r) CK(hipGraphLaunch(ge, s));
    hipEventRecord(b, s); hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("graph replay touch x500:  %.2f us/kernel\n", ms * 1000.0 / 2000);
    return 0;
}
