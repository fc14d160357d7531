// Launch descriptors shared by the encoder-stack kernels (stack_kernels.hip) and the stream
// scheduler (encoder_stack.hip).  See encoder_stack.hip for the schedule these describe.
#pragma once
#include "common.hpp"

constexpr int ED_STACK_MAX_SLOTS = 8;   // layer-steps (and norms) that ride in one launch

// interleaved gate-column order of the stack's G / dG matrices: 64-column groups
// [16 units x (i,f,g,o)], so the 4 gates of 16 consecutive units are one 128-byte line
__host__ __device__ inline int ed_gate_col(int gate, int j) {
    return (j >> 4) * 64 + gate * 16 + (j & 15);
}

struct EdFwdStep {             // one LSTM time step of one layer, all batch rows
    bf16_t* G_t;               // [B, 4H] interleaved: in pre-activations, out gates i,f,g,o
    const bf16_t* hfrag_in;    // h_{t-1} as MFMA A-fragment image [H/32][B16/16][64][8] (k-step major)
    bf16_t* hfrag_out;         // h_t, same layout (ping-pong partner)
    bf16_t* Y_t;               // [B, H] plain h_t
    const float* C_prev;       // [B, H] c_{t-1}
    float* C_t;                // [B, H] c_t
    const bf16_t* Wfrag;       // W_hh B-fragment image [H/16][H/32][4 gates][64][8]
    const unsigned* wait_flag; // null, or: G_t's chunk product (side stream) is done when *wait_flag != 0
};

struct EdFwdNorm {             // LayerNorm(y + r) of one frame, or the pair mean of two frames
    const bf16_t* y0;          // [B, H]
    const bf16_t* r0;          // [B, H] residual or null
    const bf16_t* y1;          // second frame of a time-reduction pair, or null
    const bf16_t* r1;
    const float* gamma;
    const float* beta;
    bf16_t* out;               // row b at out + b * out_stride
    long long out_stride;
    float* mean0; float* rstd0;   // [B] saved statistics of frame 0 / frame 1
    float* mean1; float* rstd1;
    float scale;               // 1 (no reduction) or 0.5 (pair mean; a missing partner counts as 0)
    // nn.Dropout behind the LayerNorm (+ TimeReduction) of the layer (rnnt/models.py:47-53,70), training only: element
    // (b, tau, j) of the layer's [B, drop_T, H] output is kept iff ed_drop_keep(drop_seed, (b drop_T + tau) H + j,
    // drop_thresh) and scaled by drop_scale = 1 / (1 - p); drop_thresh = 0: no dropout
    unsigned drop_thresh, drop_seed;
    float drop_scale;
    int drop_T, tau;
};

struct EdFwdLaunch {
    EdFwdStep step[ED_STACK_MAX_SLOTS];
    EdFwdNorm norm[ED_STACK_MAX_SLOTS];
    int nstep, nnorm;
    int B, H;
    float eps;
    unsigned long long* stamp;   // measurement mode only (else null): [0] min start, [1] max end, 100 MHz ticks
    unsigned* err;               // host-visible give-up word of the bounded flag waits (may be null)
};

struct EdBwdStep {             // one BPTT step of one layer, all batch rows
    bf16_t* G_t;               // [B, 4H] interleaved: in gates, out dL/d(pre-activation)
    const bf16_t* gfrag_in;    // dG_{t+1} A-fragment image [4H/32][B16/16][64][8]; null at t = T-1
    bf16_t* gfrag_out;         // dG_t image; null at t = 0
    const bf16_t* dY_t;        // [B, H] dL/dh_t from above (null = zeros)
    const float* C_t;          // [B, H]
    const float* C_prev;       // [B, H]
    float* dC;                 // [B, H] running dL/dc, in/out
    const bf16_t* WTfrag;      // W_hh^T B-fragment image [H/32][4H/32][2][64][8], K interleaved
    const unsigned* wait_flag; // null, or: dY_t's chunk (dX product + LayerNorm backward on the side stream) is
                               // done when *wait_flag != 0
};

struct EdBwdLaunch {
    EdBwdStep step[ED_STACK_MAX_SLOTS];
    int nstep;
    int B, H;
    unsigned long long* stamp;   // as EdFwdLaunch::stamp
    unsigned* err;               // as EdFwdLaunch::err
};

// ---- launch-persistent wavefront (stack_kernels.hip, stack_fwd_lpw_kernel): ONE launch carries `nsteps`
// CONSECUTIVE time steps of every runnable layer.  A workgroup keeps its W_hh slice in registers and its
// cell state in LDS for the whole launch; between two steps the layer's workgroups meet through an arrival
// counter and exchange h through the fragment images with write-through stores / L2-served loads.
struct EdLpwSlot {
    bf16_t* G;                 // step t0: [B, 4H] interleaved (in pre-activations, out gates); step t0+s at + s*B*4H
    bf16_t* img;               // h fragment images [T+1][H/32][B16/16][64][8], ONE PER FRAME (data polling: images 1 .. T hold
                               // all ones before the pass - a reader recognises what is not written yet): step t reads image t
                               // (h_{t-1}), writes image t+1.  No address is written twice inside a forward
                               // pass, so no cache can hold a stale copy: readers use plain loads and the
                               // 8 workgroups of a layer that share an XCD fetch an image over the fabric once
                               // (with L2-bypassing loads of ping-pong images every workgroup did: 32 MB per step)
    long long img_stride;      // bytes between consecutive images
    long long img_bytes;       // bytes of the whole region (buffer descriptor)
    bf16_t* Y;                 // h rows of step t0 [B, H]; step t0+s at + s*B*H
    const float* C_prev;       // c_{t0-1} [B, H]
    float* C;                  // c rows of step t0; step t0+s at + s*B*H
    const bf16_t* Wfrag;       // W_hh B-fragment image (EdFwdStep::Wfrag)
    unsigned* counter;         // arrivals of this layer's workgroups, one per finished step (zeroed per call).  Data polling:
                               // only the side streams read it (stack_wait_counters_kernel), and the arrival of a step
                               // comes one step late
    unsigned base;             // *counter once every step < t0 is done = workgroups_per_step * (steps done before)
    const unsigned* wait_flag; // null, or: step t0 opens a chunk whose side-stream product is done when != 0
    int t0, nsteps;
    int layer;                 // for the debug trace only
};
constexpr int LPW_CNT_STRIDE = 64;   // words between arrival counters: each has its own 256-byte line (the four layers of a
                                     // launch are polled by 256 lanes and bumped 256 times per step - in ONE line they queued
                                     // in one memory channel: 4-slot launches 80-127 us per 6 steps instead of 50-56)
struct EdLpwLaunch {
    EdLpwSlot slot[ED_STACK_MAX_SLOTS];
    int nslot;
    int data_poll;               // 1: the images were filled with all-ones before the pass and the readers validate what they
                                 // gather (no counter on the dependency chain); 0: the readers poll the arrival counters
    int B, H;
    unsigned long long* stamp;   // as EdFwdLaunch::stamp
    unsigned* err;               // host-visible give-up word (may be null)
    long long* trace;            // debug (nullable): per-slot phase times of workgroup 0, see tools/lpw_trace.py
};
int ed_stack_launch_fwd_lpw(const EdLpwLaunch& L, hipStream_t s);
// LayerNorm (+ residual, + pair mean under time reduction) of frames [t0, t1) of up to 8 layers in one launch
struct EdChunkNorm {
    const bf16_t* Yx1;         // h_t at Yx1 + t * B * H
    const bf16_t* X;           // residual rows (time-major) or null
    const float* gamma;
    const float* beta;
    bf16_t* out;               // frame tau, row b at out + tau * out_st + b * out_sb
    long long out_st, out_sb;
    float* mean;
    float* rstd;
    int T, t0, t1, reduce;
    unsigned drop_thresh, drop_seed;   // as EdFwdNorm (drop_thresh = 0: no dropout); the layer's output has
    float drop_scale;                  // drop_T = ceil(T / reduce) frames
    int drop_T;
};
// one lane spins on stream s until counters[i] >= targets[i] for every i (bounded: give-up code 800 + i)
int ed_stack_wait_counters(const unsigned* const* counters, const unsigned* targets, int n, unsigned* err, hipStream_t s);
int ed_stack_multi_norm(const EdChunkNorm* items, int n, int B, int H, float eps, hipStream_t s);
int ed_stack_lpw_supported(int B, int H);     // 1 when the launch-persistent forward kernel covers this geometry

// ---- split-K, weights-stationary BPTT (stack_kernels.hip, stack_bwd_sk_kernel; needs B <= 64, H % 64 == 0,
// H <= 1024).  The launch-per-step BPTT moves 512 KB into every CU per step (W_hh^T slice 256 KB + dG image
// 256 KB) and a CU pulls 40-60 GB/s: 13-17 us.  Here a workgroup owns (64 units, one QUARTER of the 4H gate
// columns) for all 64 rows: its W_hh^T slice is 128 KB and stays in 128 registers per lane for the whole launch,
// the only dependent fetch of a step is its quarter of the dG image (128 KB).  The four workgroups of a unit
// block exchange their partial sums (16 KB fp32 each, write-through) and each finishes 16 of the 64 rows.
constexpr int SK_CNT_QUARTER = 16;    // first quarter-counter line of a layer's gcounter block (behind the 16 unit blocks)
constexpr int SK_CNT_LINES = 32;      // 256-byte counter lines per layer: unit-block counters [0, 16), quarter counters [16, 20)
struct EdSkSlot {
    bf16_t* G;                 // frame t0 [B, 4H] interleaved (in gates, out dL/d(pre-activation)); frame t0-s at - s*B*4H
    bf16_t* img;               // dG fragment images [T + 1][4H/32][B16/16][64][8], one per frame: step t reads image t+1
                               // (absent at t = T-1), writes image t (not at t = 0)
    long long img_stride, img_bytes;
    const bf16_t* dY;          // dL/dh rows of frame t0 from above [B, H]; frame t0-s at - s*B*H
    const float* Cx;           // c_t at Cx + (t+1)*B*H
    float* dC;                 // [B, H] running dL/dc, in/out
    const bf16_t* Wsk;         // split-K fragment image of W_hh (edgedict_stack_pack_sk)
    float* part;               // [2][H/64][4][64][64] f32 partial sums, ping-pong by step parity
    unsigned* counter;         // H % 256 != 0: the layer's workgroups, one arrival per finished step
    unsigned* gcounter;        // lines of 64 words: [ub] the 4 workgroups of unit block ub, one arrival each per step once
                               // their partial is in memory; [SK_CNT_QUARTER + q] (H % 256 == 0) the workgroups that write
                               // quarter q of the gate columns, one arrival per step
    unsigned done;             // BPTT steps of this layer done before this launch (every counter is a multiple of it)
    const unsigned* wait_flag;
    int t0, nsteps, T, layer;
};
struct EdSkLaunch {
    EdSkSlot slot[ED_STACK_MAX_SLOTS];
    int nslot;
    int B, H;
    unsigned long long* stamp;
    unsigned* err;
    long long* trace;          // debug (nullable): per-slot phase times of workgroup 0, see tools/sk_trace.py
};
int ed_stack_launch_bwd_sk(const EdSkLaunch& L, hipStream_t s);
int ed_stack_sk_supported(int B, int H);
int ed_stack_pack_sk(const float* w_hh, bf16_t* out, int H, hipStream_t s);

// kernels / launchers implemented in stack_kernels.hip
int ed_stack_launch_fwd(const EdFwdLaunch& L, hipStream_t s);
int ed_stack_set_flag(unsigned* flag, hipStream_t s);   // *flag = 1 once the stream reaches this point
int ed_stack_launch_bwd(const EdBwdLaunch& L, hipStream_t s);
// time-major LayerNorm backward over frames [t0, t1) of one layer; workgroup j writes its
// dgamma/dbeta partial sums to part[j][2][H] (grid rows), summed later by ed_stack_sum_parts
int ed_stack_ln_bwd(const bf16_t* dout, long long dout_st, long long dout_sb, const bf16_t* y,
                    const bf16_t* res, const float* gamma, const float* mean, const float* rstd,
                    bf16_t* dz, float* part, int grid, int B, int H, int t0, int t1, int reduce,
                    hipStream_t s, unsigned drop_thresh = 0, unsigned drop_seed = 0, float drop_scale = 1.f,
                    int drop_T = 0);
int ed_stack_sum_parts(const float* part, int rows, int H, float* dgamma, float* dbeta, hipStream_t s);
int ed_stack_input_norm(int x_dtype, const void* x, const float* gamma, const float* beta,
                        bf16_t* out, float* mean, float* rstd, int B, int T, int D, float eps,
                        hipStream_t s);
int ed_stack_input_norm_bwd(int x_dtype, const void* x, const bf16_t* dX, const float* mean,
                            const float* rstd, float* dgamma, float* dbeta, int B, int T, int D,
                            hipStream_t s);
struct EdInitStates {      // initial states of up to 8 layers, one launch
    int n, B, H;
    const float* h0[8];
    const float* c0[8];
    bf16_t* Yx0[8];
    float* Cx0[8];
    bf16_t* hfrag[8];
};
int ed_stack_init_states(const EdInitStates& A, hipStream_t s);
int ed_stack_init_state(const float* h0, const float* c0, bf16_t* Yx0, float* Cx0, bf16_t* hfrag,
                        int B, int H, hipStream_t s);
int ed_stack_zero(void* p, size_t bytes, hipStream_t s);
