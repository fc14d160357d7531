// Probe: where does the time of one LSTM step kernel go?  Build variants with -DED_LSTM_DBG=<mask>
//   bit0 skip fragment loads + MFMA   bit1 skip gate stores to G   bit2 skip Cst/Y/Hprev/hfrag stores
//   bit3 replace expf/tanhf by cheap arithmetic   bit4 skip early G/c loads
#include "../edgedict_amd/csrc/lstm_fast.hip"
#include "../edgedict_amd/csrc/error.cpp"
#include <vector>

int main(int argc, char** argv) {
    const int B = 64, T = 401, H = 1024;
    hipStream_t s; hipStreamCreate(&s);
    void *G, *Hprev, *Y, *Wf, *Wb, *ws, *dY; float *Cst, *dC;
    hipMalloc(&G, (size_t)B * T * 4 * H * 2); hipMalloc(&Hprev, (size_t)B * T * H * 2);
    hipMalloc(&Y, (size_t)B * T * H * 2); hipMalloc(&dY, (size_t)B * T * H * 2);
    hipMalloc(&Cst, (size_t)B * T * H * 4);
    hipMalloc(&Wf, (size_t)4 * H * H * 2); hipMalloc(&Wb, (size_t)4 * H * H * 2);
    hipMalloc(&ws, ed_lstm_fast_ws_bytes(B, H)); hipMalloc(&dC, (size_t)B * H * 4);
    hipMemset(G, 0, (size_t)B * T * 4 * H * 2); hipMemset(Wf, 0, (size_t)4 * H * H * 2);
    hipMemset(Wb, 0, (size_t)4 * H * H * 2); hipMemset(dY, 0, (size_t)B * T * H * 2);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a, s);
        ed_lstm_fwd_fast(G, Hprev, Y, Cst, Wf, nullptr, nullptr, nullptr, nullptr, B, T, H, ws, s);
        hipEventRecord(b, s); hipStreamSynchronize(s);
        hipEventElapsedTime(&ms, a, b);
        if (rep) printf("DBG=%d fwd eager: %.2f us/step\n", ED_LSTM_DBG, ms * 1000 / T);
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a, s);
        ed_lstm_bwd_fast(G, dY, Cst, nullptr, Wb, dC, B, T, H, ws, s);
        hipEventRecord(b, s); hipStreamSynchronize(s);
        hipEventElapsedTime(&ms, a, b);
        if (rep) printf("DBG=%d bwd eager: %.2f us/step\n", ED_LSTM_DBG, ms * 1000 / T);
    }
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    ed_lstm_fwd_fast(G, Hprev, Y, Cst, Wf, nullptr, nullptr, nullptr, nullptr, B, T, H, ws, s);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(a, s); hipGraphLaunch(ge, s); hipEventRecord(b, s); hipStreamSynchronize(s);
    hipEventElapsedTime(&ms, a, b);
    printf("DBG=%d fwd graph: %.2f us/step\n", ED_LSTM_DBG, ms * 1000 / T);
    // 4 independent row-tile chains on 4 streams, captured into one graph (fork / join by events)
    for (int ns : {1, 2, 4}) {
        hipStream_t st[4]; hipEvent_t fork, join[4];
        for (int i = 0; i < 4; ++i) { hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking); hipEventCreateWithFlags(&join[i], hipEventDisableTiming); }
        hipEventCreateWithFlags(&fork, hipEventDisableTiming);
        hipGraph_t g2; hipGraphExec_t ge2;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        hipEventRecord(fork, s);
        for (int i = 0; i < ns; ++i) hipStreamWaitEvent(st[i], fork, 0);
        ed_lstm_fwd_fast_chains(G, Hprev, Y, Cst, Wf, nullptr, nullptr, nullptr, B, T, H, ws, st, ns);
        for (int i = 0; i < ns; ++i) { hipEventRecord(join[i], st[i]); hipStreamWaitEvent(s, join[i], 0); }
        hipError_t e = hipStreamEndCapture(s, &g2);
        if (e != hipSuccess) { printf("capture failed %s\n", hipGetErrorString(e)); return 1; }
        hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0);
        hipGraphLaunch(ge2, s); hipStreamSynchronize(s);
        hipEventRecord(a, s); hipGraphLaunch(ge2, s); hipEventRecord(b, s); hipStreamSynchronize(s);
        hipEventElapsedTime(&ms, a, b);
        printf("DBG=%d fwd graph, %d stream chains: %.2f us per full step (T=%d)\n", ED_LSTM_DBG, ns, ms * 1000 / T, T);
    }
    return 0;
}
