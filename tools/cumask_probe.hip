// Probe: does a CU-masked stream (hipExtStreamCreateWithCUMask) confine a kernel's workgroups to chosen
// XCDs, so that ordinary kernels can run BESIDE a persistent kernel that fills the other XCDs?
// (Without a mask a grid's workgroups are bound round-robin to all 8 XCDs at dispatch: every second
// kernel stalls behind a grid that fills whole XCDs - measured with tools/wsr_trace.py.)
//   1. census: for several mask layouts, which XCC ids do the workgroups of a 2048-block grid report?
//   2. concurrency: a 140 KB-LDS "persistent" kernel holds XCDs 0..5 for ~2 ms; how long does a
//      256-block kernel on (a) an ordinary stream, (b) the masked stream take to finish?
//   hipcc -O3 --offload-arch=gfx950 tools/cumask_probe.hip -o tools/cumask_probe.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) {                                               \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));    \
            exit(2);                                                          \
        }                                                                     \
    } while (0)

__global__ void census(unsigned* cnt, int spin) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) atomicAdd(&cnt[xcc & 7], 1u);
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
}

__global__ __launch_bounds__(256, 1) void hog(unsigned* cnt, long long ticks) {
    extern __shared__ unsigned char lds[];
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    if (xcc >= 6) return;
    lds[threadIdx.x] = 1;
    if (threadIdx.x == 0) atomicAdd(&cnt[8 + xcc], 1u);
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    unsigned* cnt;
    CK(hipMalloc(&cnt, 256));
    const char* names[] = {"bits 192..255", "bits i%8 in {6,7}", "bits 0..63", "bits (i/4)%8 in {6,7}", "bits (i/2)%8 in {6,7}", "bits (i/16)%8>=6"};
    hipStream_t masked[6];
    for (int v = 0; v < 6; ++v) {
        uint32_t mask[8];
        memset(mask, 0, sizeof(mask));
        for (int i = 0; i < 256; ++i) {
            bool on = false;
            if (v == 0) on = i >= 192;
            if (v == 1) on = (i % 8) >= 6;
            if (v == 2) on = i < 64;
            if (v == 3) on = ((i / 4) % 8) >= 6;
            if (v == 4) on = ((i / 2) % 8) >= 6;
            if (v == 5) on = ((i / 16) % 8) >= 6;
            if (on) mask[i / 32] |= 1u << (i % 32);
        }
        hipError_t e = hipExtStreamCreateWithCUMask(&masked[v], 8, mask);
        if (e != hipSuccess) {
            printf("%-24s: hipExtStreamCreateWithCUMask -> %s\n", names[v], hipGetErrorString(e));
            masked[v] = nullptr;
            continue;
        }
        CK(hipMemset(cnt, 0, 256));
        hipLaunchKernelGGL(census, dim3(2048), dim3(64), 0, masked[v], cnt, 2000);
        CK(hipStreamSynchronize(masked[v]));
        unsigned h[64];
        CK(hipMemcpy(h, cnt, 256, hipMemcpyDeviceToHost));
        printf("%-24s: workgroups per XCC:", names[v]);
        for (int x = 0; x < 8; ++x) printf(" %4u", h[x]);
        printf("\n");
    }
    // ---- concurrency beside a grid that fills XCDs 0..5
    hipStream_t P, S;
    CK(hipStreamCreateWithFlags(&P, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
    CK(hipFuncSetAttribute((const void*)hog, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    for (int v = -1; v < 6; ++v) {
        hipStream_t side = v < 0 ? S : masked[v];
        if (!side) continue;
        CK(hipMemset(cnt, 0, 256));
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(hog, dim3(256), dim3(256), 140 * 1024, P, cnt, 200000);   // 2 ms
        CK(hipEventRecord(e0, side));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(census, dim3(256), dim3(256), 0, side, cnt, 1000);
        CK(hipEventRecord(e1, side));
        CK(hipEventRecord(e2, P));
        CK(hipDeviceSynchronize());
        float ms_side = 0, ms_all = 0;
        CK(hipEventElapsedTime(&ms_side, e0, e1));
        CK(hipEventElapsedTime(&ms_all, e0, e2));
        unsigned h[64];
        CK(hipMemcpy(h, cnt, 256, hipMemcpyDeviceToHost));
        printf("beside the hog: %-22s 10 side kernels %.3f ms (hog done at %.3f ms); side workgroups per XCC:",
               v < 0 ? "ordinary stream" : names[v], ms_side, ms_all);
        for (int x = 0; x < 8; ++x) printf(" %u", h[x]);
        printf("\n");
    }
    return 0;
}
