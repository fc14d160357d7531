// Probe: do INDEPENDENT recurrence chains (different layers: own weights, own buffers) overlap when
// they are launched on separate HIP streams?  This is the question behind the layer-pipelined
// (chunked wavefront) encoder schedule.  Reports total time / T for L concurrent chains:
//   mode rr     one host thread, launches interleaved round-robin over L streams
//   mode thr    one host thread per stream
//   mode graph  the L-stream fork/join captured once and replayed
// hipcc --offload-arch=gfx950 -O3 tools/pipeline_probe.hip -o tools/pipeline_probe.bin
#include "../edgedict_amd/csrc/lstm_fast.hip"
#include "../edgedict_amd/csrc/error.cpp"
#include <chrono>
#include <thread>
#include <vector>

struct Chain {
    void *G, *Hprev, *Y, *Wf, *ws; float* Cst; hipStream_t s;
};

static void launch_step(const Chain& c, int B, int T, int H, int t) {
    const size_t half = (frag_bytes(B, 4 * H) + 255) / 256 * 256;
    bf16_t* frag[2] = {(bf16_t*)c.ws, (bf16_t*)((char*)c.ws + half)};
    hipLaunchKernelGGL(lstm_step_fwd_fast, dim3(H / 16, (B + 15) / 16), dim3(256), 0, c.s,
                       (bf16_t*)c.G, frag[t & 1], frag[(t + 1) & 1], (bf16_t*)c.Hprev, (bf16_t*)c.Y,
                       c.Cst, (const bf16_t*)c.Wf, nullptr, nullptr, nullptr, B, T, H, t, 0);
}

int main() {
    const int B = 64, T = 201, H = 1024, LMAX = 6;
    std::vector<Chain> ch(LMAX);
    for (auto& c : ch) {
        hipMalloc(&c.G, (size_t)B * T * 4 * H * 2); hipMalloc(&c.Hprev, (size_t)B * T * H * 2);
        hipMalloc(&c.Y, (size_t)B * T * H * 2); hipMalloc(&c.Cst, (size_t)B * T * H * 4);
        hipMalloc(&c.Wf, (size_t)4 * H * H * 2); hipMalloc(&c.ws, ed_lstm_fast_ws_bytes(B, H));
        hipMemset(c.G, 0, (size_t)B * T * 4 * H * 2); hipMemset(c.Wf, 0, (size_t)4 * H * H * 2);
        hipMemset(c.ws, 0, ed_lstm_fast_ws_bytes(B, H));
        hipStreamCreateWithFlags(&c.s, hipStreamNonBlocking);
    }
    hipStream_t s0; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    hipEvent_t a, b, fork, join[LMAX];
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventCreateWithFlags(&fork, hipEventDisableTiming);
    for (auto& e : join) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    float ms;
    for (int L : {1, 2, 3, 4, 6}) {
        for (int rep = 0; rep < 2; ++rep) {   // ---- rr
            hipDeviceSynchronize();
            auto t0 = std::chrono::high_resolution_clock::now();
            hipEventRecord(a, s0); hipEventRecord(fork, s0);
            for (int l = 0; l < L; ++l) hipStreamWaitEvent(ch[l].s, fork, 0);
            for (int t = 0; t < T; ++t)
                for (int l = 0; l < L; ++l) launch_step(ch[l], B, T, H, t);
            for (int l = 0; l < L; ++l) { hipEventRecord(join[l], ch[l].s); hipStreamWaitEvent(s0, join[l], 0); }
            hipEventRecord(b, s0);
            auto t1 = std::chrono::high_resolution_clock::now();
            hipStreamSynchronize(s0);
            hipEventElapsedTime(&ms, a, b);
            if (rep) printf("L=%d rr    : %.2f us per time index (%.2f us/kernel), host enqueue %.2f us/launch\n", L,
                            ms * 1000 / T, ms * 1000 / T / L,
                            std::chrono::duration<double, std::micro>(t1 - t0).count() / (T * L));
        }
        for (int rep = 0; rep < 2; ++rep) {   // ---- threads
            hipDeviceSynchronize();
            hipEventRecord(a, s0); hipEventRecord(fork, s0);
            for (int l = 0; l < L; ++l) hipStreamWaitEvent(ch[l].s, fork, 0);
            std::vector<std::thread> th;
            for (int l = 0; l < L; ++l)
                th.emplace_back([&, l] { for (int t = 0; t < T; ++t) launch_step(ch[l], B, T, H, t); });
            for (auto& x : th) x.join();
            for (int l = 0; l < L; ++l) { hipEventRecord(join[l], ch[l].s); hipStreamWaitEvent(s0, join[l], 0); }
            hipEventRecord(b, s0);
            hipStreamSynchronize(s0);
            hipEventElapsedTime(&ms, a, b);
            if (rep) printf("L=%d thr   : %.2f us per time index (%.2f us/kernel)\n", L, ms * 1000 / T, ms * 1000 / T / L);
        }
        {   // ---- graph
            hipGraph_t g; hipGraphExec_t ge;
            hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal);
            hipEventRecord(fork, s0);
            for (int l = 0; l < L; ++l) hipStreamWaitEvent(ch[l].s, fork, 0);
            for (int t = 0; t < T; ++t)
                for (int l = 0; l < L; ++l) launch_step(ch[l], B, T, H, t);
            for (int l = 0; l < L; ++l) { hipEventRecord(join[l], ch[l].s); hipStreamWaitEvent(s0, join[l], 0); }
            if (hipStreamEndCapture(s0, &g) != hipSuccess) { printf("capture failed\n"); return 1; }
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            hipGraphLaunch(ge, s0); hipStreamSynchronize(s0);
            auto t0 = std::chrono::high_resolution_clock::now();
            hipEventRecord(a, s0); hipGraphLaunch(ge, s0); hipEventRecord(b, s0);
            auto t1 = std::chrono::high_resolution_clock::now();
            hipStreamSynchronize(s0);
            hipEventElapsedTime(&ms, a, b);
            printf("L=%d graph : %.2f us per time index (%.2f us/kernel), host %.0f us total\n", L, ms * 1000 / T,
                   ms * 1000 / T / L, std::chrono::duration<double, std::micro>(t1 - t0).count());
            hipGraphExecDestroy(ge); hipGraphDestroy(g);
        }
    }
    return 0;
}
