// Probe: can a RUNNING persistent kernel hand work to / take results from ordinary kernels on another
// stream without any device-side polling kernel?  (hipStreamWaitValue32 / hipStreamWriteValue32 are
// executed by the command processor: they hold no CU, so they cannot starve a persistent kernel
// whose workgroups need whole CUs.)
//
//   kernel P (persistent, one 140 KB-LDS workgroup per CU on XCDs 0..5; XCD 6,7 workgroups exit):
//       for i in 0..N:  X = i+1 (system-scope store)  ->  poll G >= i+1 (bounded)  -> stamp latency
//   stream S:  for i in 0..N:  WaitValue32(X >= i+1) -> kernel K (a stand-in for a chunk GEMM:
//       128 workgroups x 64 KB LDS) -> WriteValue32(G, i+1)
// Reports the round-trip time per iteration, where K's workgroups ran (XCC ids), and give-up codes.
//
//   hipcc -O3 --offload-arch=gfx950 tools/streamop_probe.hip -o tools/streamop_probe.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) {                                               \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));    \
            exit(2);                                                          \
        }                                                                     \
    } while (0)

typedef __attribute__((address_space(1))) unsigned gu32;

__global__ __launch_bounds__(256, 1) void persistent(unsigned* X, unsigned* G, unsigned* err, long long* lat,
                                                     unsigned* census, int N) {
    extern __shared__ unsigned char lds[];
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    if (threadIdx.x == 0) atomicAdd(&census[xcc], 1u);
    if (xcc >= 6) return;
    lds[threadIdx.x] = 1;   // touch the LDS so the allocation is real
    __shared__ unsigned leader_s;
    if (threadIdx.x == 0) leader_s = atomicAdd(&census[8], 1u);
    __syncthreads();
    const bool leader = leader_s == 0;
    for (int i = 0; i < N; ++i) {
        const long long t0 = wall_clock64();
        if (leader && threadIdx.x == 0)
            __hip_atomic_store(X, (unsigned)(i + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned)(i + 1)) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 17)) {
                    atomicExch(err, 1000u + (unsigned)i);
                    break;
                }
                if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            }
        }
        __syncthreads();
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
        if (leader && threadIdx.x == 0) lat[i] = wall_clock64() - t0;
    }
}

__global__ __launch_bounds__(256) void standin(float* buf, unsigned* where, int iters) {
    __shared__ float s[16384];   // 64 KB, like gemm_nt_kernel
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) atomicAdd(&where[xcc & 7], 1u);
    float v = buf[blockIdx.x * 256 + threadIdx.x];
    for (int i = 0; i < iters; ++i) {
        s[(threadIdx.x * 17 + i) & 16383] = v;
        v = v * 1.0001f + s[(threadIdx.x * 31 + i) & 16383];
    }
    buf[blockIdx.x * 256 + threadIdx.x] = v;
}

int main(int argc, char** argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    const int N = argc > 1 ? atoi(argv[1]) : 200;
    const int kind = argc > 2 ? atoi(argv[2]) : 0;   // 0 signal memory, 1 plain hipMalloc, 2 pinned host
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d, flag memory kind %d\n", can, kind);
    unsigned *X = nullptr, *G = nullptr;
    if (kind == 0) {
        hipError_t e = hipExtMallocWithFlags((void**)&X, 8, hipMallocSignalMemory);
        printf("hipExtMallocWithFlags(signal) -> %s\n", hipGetErrorString(e));
        if (e != hipSuccess) return 3;
        CK(hipExtMallocWithFlags((void**)&G, 8, hipMallocSignalMemory));
    } else if (kind == 1) {
        CK(hipMalloc(&X, 64));
        CK(hipMalloc(&G, 64));
    } else {
        CK(hipHostMalloc(&X, 64, hipHostMallocMapped));
        CK(hipHostMalloc(&G, 64, hipHostMallocMapped));
    }
    CK(hipMemset(X, 0, 8));
    CK(hipMemset(G, 0, 8));
    unsigned *err, *census, *where;
    long long* lat;
    float* buf;
    CK(hipMalloc(&err, 64));
    CK(hipMalloc(&census, 64));
    CK(hipMalloc(&where, 64));
    CK(hipMalloc(&lat, sizeof(long long) * N));
    CK(hipMalloc(&buf, 128 * 256 * 4));
    CK(hipMemset(err, 0, 64));
    CK(hipMemset(census, 0, 64));
    CK(hipMemset(where, 0, 64));
    CK(hipMemset(buf, 0, 128 * 256 * 4));
    hipStream_t P, S;
    CK(hipStreamCreateWithFlags(&P, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
    CK(hipFuncSetAttribute((const void*)persistent, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    CK(hipDeviceSynchronize());

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, P));
    hipLaunchKernelGGL(persistent, dim3(256), dim3(256), 140 * 1024, P, X, G, err, lat, census, N);
    CK(hipGetLastError());
    CK(hipEventRecord(e1, P));
    printf("persistent kernel enqueued\n");
    for (int i = 0; i < N; ++i) {
        if (i < 3) printf("enqueue wait/kernel/write %d\n", i);
        hipError_t e = hipStreamWaitValue32(S, X, (uint32_t)(i + 1), hipStreamWaitValueGte, 0xFFFFFFFFu);
        if (e != hipSuccess) {
            printf("hipStreamWaitValue32 -> %s (iteration %d)\n", hipGetErrorString(e), i);
            // release the persistent kernel: it gives up by itself after its bounded spin
            break;
        }
        hipLaunchKernelGGL(standin, dim3(128), dim3(256), 0, S, buf, where, 200);
        e = hipStreamWriteValue32(S, G, (uint32_t)(i + 1), 0);
        if (e != hipSuccess) {
            printf("hipStreamWriteValue32 -> %s\n", hipGetErrorString(e));
            break;
        }
    }
    printf("all enqueued; syncing P\n");
    CK(hipStreamSynchronize(P));
    printf("P done; syncing S\n");
    if (hipStreamQuery(S) != hipSuccess) {
        // the persistent kernel gave up: release the waits on S from the host so that it drains
        printf("S still pending: writing X from the host to release it\n");
        unsigned big = 0x7fffffff;
        hipMemcpy(X, &big, 4, hipMemcpyHostToDevice);
    }
    CK(hipStreamSynchronize(S));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned herr[16], hc[16], hw[16];
    CK(hipMemcpy(herr, err, 64, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hc, census, 64, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hw, where, 64, hipMemcpyDeviceToHost));
    long long* hl = (long long*)malloc(sizeof(long long) * N);
    CK(hipMemcpy(hl, lat, sizeof(long long) * N, hipMemcpyDeviceToHost));
    double sum = 0;
    long long mn = 1ll << 60, mx = 0;
    for (int i = 5; i < N; ++i) {
        sum += hl[i];
        if (hl[i] < mn) mn = hl[i];
        if (hl[i] > mx) mx = hl[i];
    }
    printf("persistent kernel %.3f ms for %d round trips: %.2f us each (device clock: mean %.2f min %.2f max %.2f us), err=%u\n",
           ms, N, 1e3 * ms / N, sum / (N - 5) / 100.0, mn / 100.0, mx / 100.0, herr[0]);
    printf("persistent census per XCC:");
    for (int i = 0; i < 8; ++i) printf(" %u", hc[i]);
    printf("   stand-in workgroups per XCC:");
    for (int i = 0; i < 8; ++i) printf(" %u", hw[i]);
    printf("\n");
    return 0;
}
