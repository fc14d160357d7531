// Probe: cost structure of stack_fwd_kernel (the multi-layer wavefront step).  One chain of 200
// dependent launches per configuration; NS layer-slots per launch, each slot with its own weight
// image and buffers (or all slots sharing ONE weight image: "sharedW", L2-resident) so the L2
// capacity effect of streaming 4 x 8 MB of W_hh per launch is visible.
//   hipcc --offload-arch=gfx950 -O3 -I include -I edgedict_amd/csrc -DED_STACK_DBG=<mask> tools/stack_probe.hip
#include "../edgedict_amd/csrc/stack_kernels.hip"
#include "../edgedict_amd/csrc/error.cpp"
#include <vector>

int main() {
    const int B = 64, H = 1024, T = 200, NSMAX = 8;
    struct Slot { bf16_t *G, *f0, *f1, *Y, *W; float* C; };
    std::vector<Slot> sl(NSMAX);
    for (auto& s : sl) {
        hipMalloc(&s.G, (size_t)T * B * 4 * H * 2); hipMemset(s.G, 0, (size_t)T * B * 4 * H * 2);
        hipMalloc(&s.f0, (size_t)B * H * 2); hipMalloc(&s.f1, (size_t)B * H * 2);
        hipMemset(s.f0, 0, (size_t)B * H * 2); hipMemset(s.f1, 0, (size_t)B * H * 2);
        hipMalloc(&s.Y, (size_t)(T + 1) * B * H * 2); hipMalloc(&s.C, (size_t)(T + 1) * B * H * 4);
        hipMemset(s.C, 0, (size_t)(T + 1) * B * H * 4);
        hipMalloc(&s.W, (size_t)4 * H * H * 2); hipMemset(s.W, 0, (size_t)4 * H * H * 2);
    }
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int shared = 0; shared < 2; ++shared)
        for (int ns : {1, 2, 3, 4, 5, 6, 8}) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a, st);
                for (int t = 0; t < T; ++t) {
                    EdFwdLaunch L; L.nstep = ns; L.nnorm = 0; L.B = B; L.H = H; L.eps = 1e-5f; L.stamp = nullptr; L.err = nullptr;
                    for (int i = 0; i < ns; ++i) {
                        EdFwdStep& s = L.step[i];
                        s.G_t = sl[i].G + (size_t)t * B * 4 * H;
                        s.hfrag_in = (t & 1) ? sl[i].f1 : sl[i].f0;
                        s.hfrag_out = (t & 1) ? sl[i].f0 : sl[i].f1;
                        s.Y_t = sl[i].Y + (size_t)(t + 1) * B * H;
                        s.C_prev = sl[i].C + (size_t)t * B * H;
                        s.C_t = sl[i].C + (size_t)(t + 1) * B * H;
                        s.Wfrag = shared ? sl[0].W : sl[i].W;
                        s.wait_flag = nullptr;
                    }
                    ed_stack_launch_fwd(L, st);
                }
                hipEventRecord(b, st); hipStreamSynchronize(st);
                hipEventElapsedTime(&ms, a, b);
            }
            printf("DBG=%d %s slots=%d: %.2f us/launch\n", ED_STACK_DBG, shared ? "sharedW" : "ownW   ", ns, ms * 1000 / T);
        }
    return 0;
}
