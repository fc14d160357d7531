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
// handle per device, a 64 MiB workspace per (device, stream) that has used the route, and a small
// shape -> algorithm cache.
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
        // OPT-IN since round 2 (EDGEDICT_BLASLT=1): every product of the training step has an own kernel
        // that is at least as fast inside the step (gemm_nt256 / gemm_tn256 / the ring kernel of gemm_nt.hip)
        const char* e = getenv("EDGEDICT_BLASLT");
        if (!e || e[0] == '0') return x;
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
    // one workspace PER STREAM: products on different streams run concurrently (chunk dX on the side
    // stream beside weight gradients on the low-priority stream), and stream-K kernels keep partial
    // sums and flags in the workspace - sharing one between two running kernels deadlocks them
    std::map<hipStream_t, void*> ws;
    bool failed = false;
    // (M, N, K, lda, ldb, ldc, bias) -> algorithm (valid == false: the library has none)
    struct Entry { hipblasLtMatmulAlgo_t algo; size_t ws; bool valid; };
    std::map<std::tuple<int, int, int, long long, long long, long long, int>, Entry> cache;
};

std::mutex g_mu;
PerDevice g_dev[64];
long long g_calls = 0;

}  // namespace

namespace {

// D (column-major m x n, ld ldd, type dtype_d) = alpha op(P) op(Q) + beta D (+ bias over the m rows)
// with P, Q bf16, fp32 accumulation.  Returns false when the library cannot / may not do it.
bool run(hipblasOperation_t opP, hipblasOperation_t opQ, const void* P, long long rowsP, long long colsP,
         long long ldp, const void* Q, long long rowsQ, long long colsQ, long long ldq, void* D,
         hipDataType dtype_d, int m, int n, long long ldd, float beta, const float* bias, int tag,
         hipStream_t s) {
    const Api& L = api();
    if (!L.ok) return false;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    std::lock_guard<std::mutex> lock(g_mu);
    PerDevice& d = g_dev[dev];
    if (d.failed) return false;
    if (!d.handle) {
        if (L.Create(&d.handle) != HIPBLAS_STATUS_SUCCESS) {
            d.failed = true;
            d.handle = nullptr;
            return false;
        }
    }
    void* ws = nullptr;
    {
        auto wit = d.ws.find(s);
        if (wit == d.ws.end()) {
            if (d.ws.size() >= 8 || hipMalloc(&ws, WS_BYTES) != hipSuccess) return false;   // own kernel instead
            d.ws.emplace(s, ws);
        } else {
            ws = wit->second;
        }
    }
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
    bool done = false;
    do {
        if (L.DescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) break;
        if (L.DescSet(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opP, sizeof(opP)) != HIPBLAS_STATUS_SUCCESS) break;
        if (L.DescSet(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opQ, sizeof(opQ)) != HIPBLAS_STATUS_SUCCESS) break;
        if (bias) {
            const hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_BIAS;
            const hipDataType bt = HIP_R_32F;
            if (L.DescSet(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)) != HIPBLAS_STATUS_SUCCESS) break;
            if (L.DescSet(desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)) != HIPBLAS_STATUS_SUCCESS) break;
            if (L.DescSet(desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)) != HIPBLAS_STATUS_SUCCESS) break;
        }
        if (L.LayoutCreate(&la, HIP_R_16BF, rowsP, colsP, ldp) != HIPBLAS_STATUS_SUCCESS) break;
        if (L.LayoutCreate(&lb, HIP_R_16BF, rowsQ, colsQ, ldq) != HIPBLAS_STATUS_SUCCESS) break;
        if (L.LayoutCreate(&lc, dtype_d, m, n, ldd) != HIPBLAS_STATUS_SUCCESS) break;
        const long long kdim = opP == HIPBLAS_OP_N ? colsP : rowsP;
        const auto key = std::make_tuple(m, n, (int)kdim, ldp, ldq, ldd, tag * 4 + (bias ? 1 : 0) + (beta != 0.f ? 2 : 0));
        auto it = d.cache.find(key);
        if (it == d.cache.end()) {
            PerDevice::Entry en{};
            en.valid = false;
            hipblasLtMatmulPreference_t pref = nullptr;
            if (L.PrefCreate(&pref) == HIPBLAS_STATUS_SUCCESS) {
                const uint64_t wsmax = WS_BYTES;
                L.PrefSet(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsmax, sizeof(wsmax));
                hipblasLtMatmulHeuristicResult_t res[1];
                int cnt = 0;
                if (L.Heuristic(d.handle, desc, la, lb, lc, lc, pref, 1, res, &cnt) == HIPBLAS_STATUS_SUCCESS &&
                    cnt > 0 && res[0].state == HIPBLAS_STATUS_SUCCESS && res[0].workspaceSize <= WS_BYTES) {
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
        const float one = 1.f;
        if (L.Matmul(d.handle, desc, &one, P, la, Q, lb, &beta, D, lc, D, lc, &it->second.algo, ws,
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

}  // namespace

bool ed_blaslt_nt_bf16(const void* A, long long lda, const void* B, long long ldb, void* C,
                       long long ldc, int M, int N, int K, const float* bias, int accumulate,
                       hipStream_t s) {
    // row-major C[M,N] = A[M,K] B[N,K]^T  ==  column-major D[N,M] = op(B)[N,K] A'[K,M] with
    // B seen as column-major [K,N] (ld ldb, transposed) and A as column-major [K,M] (ld lda)
    return run(HIPBLAS_OP_T, HIPBLAS_OP_N, B, K, N, ldb, A, K, M, lda, C, HIP_R_16BF, N, M, ldc,
               accumulate ? 1.f : 0.f, bias, 0, s);
}

bool ed_blaslt_tn_f32(const void* A, long long lda, const void* B, long long ldb, float* C,
                      long long ldc, int M, int N, int K, int accumulate, hipStream_t s) {
    // row-major C[M,N] (+)= A[K,M]^T B[K,N] (both operands K-strided: weight gradients)
    //   == column-major D[N,M] = B'[N,K] A'[M,K]^T, B' = B seen column-major [N,K] (ld ldb),
    //      A' = A seen column-major [M,K] (ld lda)
    // EDGEDICT_BLASLT_BG: 0 none, 1 the encoder stack's weight gradients (the "partials" callers),
    // 2 also direct accumulating products (the joint's dW2)
    static const int mode = [] {
        const char* e = getenv("EDGEDICT_BLASLT_BG");
        return e ? atoi(e) : 1;   // measured: 27.0 ms (1), 27.7 (2: the loud dW2 delays the
                                  // small critical kernels between the joint's and the stack's backward), 27.7 (0)
    }();
    if (mode <= 0 || (accumulate && mode < 2)) return false;
    return run(HIPBLAS_OP_N, HIPBLAS_OP_T, B, N, K, ldb, A, M, K, lda, C, HIP_R_32F, N, M, ldc,
               accumulate ? 1.f : 0.f, nullptr, 1, s);
}

// number of products the vendor route has taken in this process (bench.py labels its MFMA
// roofline block with the kernel that actually ran)
extern "C" long long edgedict_blaslt_calls(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    return g_calls;
}
