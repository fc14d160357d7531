// hipBLASLt for ONE kind of product: C[M,N] (bf16) = A[M,K] B[N,K]^T (+ fp32 bias[N]) with fp32
// accumulation, when it is large and output-heavy (the joint's logits product: M = 543 526 lattice
// rows, N = 2048, K = 640 - only 10 k-steps per output tile, so a tile kernel's prologue/epilogue
// is half of its time: 2.27 ms = 626 TF/s in gemm_nt.hip vs 1.43 ms = 1000 TF/s in the library,
// tools/blas_probe.py).  Every other product stays on this library's own kernels: the long-K
// joint backward product is faster there (1.50 vs 1.62 ms), the recurrence-side products are small
// and stream-ordered, the background weight-gradient products must be "quiet".
//
// The library is bound lazily with dlopen/dlsym (the process usually has PyTorch's copy loaded
// already under the same SONAME), so libedgedict_hip.so has no link-time dependency on it; if it
// is absent, or has no solution for a shape, the caller's own kernel runs.  State kept: one
// handle + a 64 MiB workspace per device and a small shape -> algorithm cache.
// EDGEDICT_BLASLT=0 turns the route off.
#include "blaslt.hpp"

#include <dlfcn.h>
#include <hipblaslt/hipblaslt.h>
#include <stdlib.h>

#include <map>
#include <mutex>
#include <tuple>

namespace {

struct Api {
    decltype(&hipblasLtCreate) Create = nullptr;
    decltype(&hipblasLtMatmulDescCreate) DescCreate = nullptr;
    decltype(&hipblasLtMatmulDescDestroy) DescDestroy = nullptr;
    decltype(&hipblasLtMatmulDescSetAttribute) DescSet = nullptr;
    decltype(&hipblasLtMatrixLayoutCreate) LayoutCreate = nullptr;
    decltype(&hipblasLtMatrixLayoutDestroy) LayoutDestroy = nullptr;
    decltype(&hipblasLtMatmulPreferenceCreate) PrefCreate = nullptr;
    decltype(&hipblasLtMatmulPreferenceDestroy) PrefDestroy = nullptr;
    decltype(&hipblasLtMatmulPreferenceSetAttribute) PrefSet = nullptr;
    decltype(&hipblasLtMatmulAlgoGetHeuristic) Heuristic = nullptr;
    decltype(&hipblasLtMatmul) Matmul = nullptr;
    bool ok = false;
};

const Api& api() {
    static const Api a = [] {
        Api x;
        const char* e = getenv("EDGEDICT_BLASLT");
        if (e && e[0] == '0') return x;
        void* h = dlopen("libhipblaslt.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libhipblaslt.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return x;
#define ED_SYM(field, name) x.field = reinterpret_cast<decltype(x.field)>(dlsym(h, name))
        ED_SYM(Create, "hipblasLtCreate");
        ED_SYM(DescCreate, "hipblasLtMatmulDescCreate");
        ED_SYM(DescDestroy, "hipblasLtMatmulDescDestroy");
        ED_SYM(DescSet, "hipblasLtMatmulDescSetAttribute");
        ED_SYM(LayoutCreate, "hipblasLtMatrixLayoutCreate");
        ED_SYM(LayoutDestroy, "hipblasLtMatrixLayoutDestroy");
        ED_SYM(PrefCreate, "hipblasLtMatmulPreferenceCreate");
        ED_SYM(PrefDestroy, "hipblasLtMatmulPreferenceDestroy");
        ED_SYM(PrefSet, "hipblasLtMatmulPreferenceSetAttribute");
        ED_SYM(Heuristic, "hipblasLtMatmulAlgoGetHeuristic");
        ED_SYM(Matmul, "hipblasLtMatmul");
#undef ED_SYM
        x.ok = x.Create && x.DescCreate && x.DescDestroy && x.DescSet && x.LayoutCreate &&
               x.LayoutDestroy && x.PrefCreate && x.PrefDestroy && x.PrefSet && x.Heuristic && x.Matmul;
        return x;
    }();
    return a;
}

constexpr size_t WS_BYTES = 64ull << 20;

struct PerDevice {
    hipblasLtHandle_t handle = nullptr;
    void* ws = nullptr;
    bool failed = false;
    // (M, N, K, lda, ldb, ldc, bias) -> algorithm (valid == false: the library has none)
    struct Entry { hipblasLtMatmulAlgo_t algo; size_t ws; bool valid; };
    std::map<std::tuple<int, int, int, long long, long long, long long, int>, Entry> cache;
};

std::mutex g_mu;
PerDevice g_dev[64];
long long g_calls = 0;

}  // namespace

bool ed_blaslt_nt_bf16(const void* A, long long lda, const void* B, long long ldb, void* C,
                       long long ldc, int M, int N, int K, const float* bias, hipStream_t s) {
    const Api& L = api();
    if (!L.ok) return false;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    std::lock_guard<std::mutex> lock(g_mu);
    PerDevice& d = g_dev[dev];
    if (d.failed) return false;
    if (!d.handle) {
        if (L.Create(&d.handle) != HIPBLAS_STATUS_SUCCESS || hipMalloc(&d.ws, WS_BYTES) != hipSuccess) {
            d.failed = true;
            d.handle = nullptr;
            return false;
        }
    }
    // row-major C[M,N] = A[M,K] B[N,K]^T  ==  column-major D[N,M] = op(B)[N,K] A'[K,M] with
    // B seen as column-major [K,N] (ld ldb, transposed) and A as column-major [K,M] (ld lda)
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
    bool done = false;
    do {
        if (L.DescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) break;
        const hipblasOperation_t opT = HIPBLAS_OP_T, opN = HIPBLAS_OP_N;
        if (L.DescSet(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opT, sizeof(opT)) != HIPBLAS_STATUS_SUCCESS) break;
        if (L.DescSet(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opN, sizeof(opN)) != HIPBLAS_STATUS_SUCCESS) break;
        if (bias) {
            const hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_BIAS;
            const hipDataType bt = HIP_R_32F;
            if (L.DescSet(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)) != HIPBLAS_STATUS_SUCCESS) break;
            if (L.DescSet(desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)) != HIPBLAS_STATUS_SUCCESS) break;
            if (L.DescSet(desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)) != HIPBLAS_STATUS_SUCCESS) break;
        }
        if (L.LayoutCreate(&la, HIP_R_16BF, K, N, ldb) != HIPBLAS_STATUS_SUCCESS) break;   // "A" = B^T source
        if (L.LayoutCreate(&lb, HIP_R_16BF, K, M, lda) != HIPBLAS_STATUS_SUCCESS) break;
        if (L.LayoutCreate(&lc, HIP_R_16BF, N, M, ldc) != HIPBLAS_STATUS_SUCCESS) break;
        const auto key = std::make_tuple(M, N, K, lda, ldb, ldc, bias ? 1 : 0);
        auto it = d.cache.find(key);
        if (it == d.cache.end()) {
            PerDevice::Entry en{};
            en.valid = false;
            hipblasLtMatmulPreference_t pref = nullptr;
            if (L.PrefCreate(&pref) == HIPBLAS_STATUS_SUCCESS) {
                const uint64_t wsmax = WS_BYTES;
                L.PrefSet(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsmax, sizeof(wsmax));
                hipblasLtMatmulHeuristicResult_t res[1];
                int n = 0;
                if (L.Heuristic(d.handle, desc, la, lb, lc, lc, pref, 1, res, &n) == HIPBLAS_STATUS_SUCCESS &&
                    n > 0 && res[0].state == HIPBLAS_STATUS_SUCCESS && res[0].workspaceSize <= WS_BYTES) {
                    en.algo = res[0].algo;
                    en.ws = res[0].workspaceSize;
                    en.valid = true;
                }
                L.PrefDestroy(pref);
            }
            if (d.cache.size() > 4096) d.cache.clear();   // lattice sizes change every batch
            it = d.cache.emplace(key, en).first;
        }
        if (!it->second.valid) break;
        const float one = 1.f, zero = 0.f;
        if (L.Matmul(d.handle, desc, &one, B, la, A, lb, &zero, C, lc, C, lc, &it->second.algo, d.ws,
                     WS_BYTES, s) != HIPBLAS_STATUS_SUCCESS) {
            it->second.valid = false;
            break;
        }
        done = true;
        ++g_calls;
    } while (0);
    if (la) L.LayoutDestroy(la);
    if (lb) L.LayoutDestroy(lb);
    if (lc) L.LayoutDestroy(lc);
    if (desc) L.DescDestroy(desc);
    return done;
}

// number of products the vendor route has taken in this process (bench.py labels its MFMA
// roofline block with the kernel that actually ran)
extern "C" long long edgedict_blaslt_calls(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    return g_calls;
}
