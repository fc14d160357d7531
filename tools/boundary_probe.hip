// Probe: what does a kernel running on ANOTHER stream do to the cost of dependent kernel boundaries
// on this stream?  Stream A: chain of 300 tiny dependent kernels.  Stream B: one long kernel of a
// given flavour, resident the whole time (512 workgroups x 256 threads, ~2 per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/boundary_probe.hip -o tools/boundary_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void tiny256(float* p) { int i = blockIdx.x * blockDim.x + threadIdx.x; p[i] += 1.f; }

// flavour: 0 spin (ALU only)  1 stream reads  2 stream writes  3 read-modify-write  4 fp32 atomics
__global__ __launch_bounds__(256) void longk(float4* buf, size_t n4, float* acc, int flavour, long long iters, int lds_dummy) {
    extern __shared__ float sh[];
    if (lds_dummy && threadIdx.x == 0) sh[0] = 1.f;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float4 a = make_float4(0, 0, 0, 0);
    for (long long it = 0; it < iters; ++it) {
        if (flavour == 0) {
#pragma unroll 16
            for (int k = 0; k < 64; ++k) a.x = a.x * 1.0001f + 0.5f;
        } else if (flavour == 1) {
            float4 v = buf[i % n4]; a.x += v.x; a.y += v.y; i += stride;
        } else if (flavour == 2) {
            buf[i % n4] = a; i += stride;
        } else if (flavour == 3) {
            float4 v = buf[i % n4]; v.x += 1.f; buf[i % n4] = v; i += stride;
        } else {
            atomicAdd(acc + ((i * 4) % (1 << 20)), 1.f); i += stride;
        }
    }
    if (a.x == 123.456f) acc[0] = a.x + a.y;
}

int main() {
    hipStream_t A, B;
    hipStreamCreateWithFlags(&A, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&B, hipStreamNonBlocking);
    float* p; hipMalloc(&p, 1 << 22); hipMemset(p, 0, 1 << 22);
    float4* buf; const size_t n4 = (size_t)1 << 28;   // 4 GiB
    hipMalloc(&buf, n4 * 16); hipMemset(buf, 0, n4 * 16);
    float* acc; hipMalloc(&acc, 4 << 20); hipMemset(acc, 0, 4 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char* names[] = {"ALU spin", "stream reads", "stream writes", "read-modify-write", "fp32 atomics"};
    const long long iters[] = {40000, 12000, 12000, 8000, 6000};
    for (int fl = -1; fl < 5; ++fl) {
        for (int lds = 0; lds < (fl < 0 ? 1 : 2); ++lds) {
            hipDeviceSynchronize();
            if (fl >= 0) hipLaunchKernelGGL(longk, dim3(512), dim3(256), lds ? 55000 : 0, B, buf, n4, acc, fl, iters[fl], lds);
            hipEventRecord(a, A);
            for (int i = 0; i < 300; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, A, p);
            hipEventRecord(b, A);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            hipEventRecord(a, A);
            for (int i = 0; i < 300; ++i) hipLaunchKernelGGL(tiny256, dim3(256), dim3(256), 0, A, p);
            hipEventRecord(b, A);
            hipEventSynchronize(b);
            float ms2; hipEventElapsedTime(&ms2, a, b);
            hipError_t q = hipStreamQuery(B);
            printf("%-18s lds_pad=%d : tiny chain %.2f us/kernel, 256-WG chain %.2f us/kernel  (long kernel %s)\n",
                   fl < 0 ? "nothing" : names[fl], lds, ms * 1000 / 300, ms2 * 1000 / 300,
                   fl < 0 ? "-" : (q == hipSuccess ? "ALREADY DONE" : "still running"));
        }
    }
    return 0;
}
