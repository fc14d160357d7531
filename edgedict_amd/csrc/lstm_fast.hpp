// Internal interface between lstm.hip (C ABI, generic path) and lstm_fast.hip (bf16 fragment-order path).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

bool ed_lstm_fast_ok(int dtype, int H);
size_t ed_lstm_fast_ws_bytes(int B, int H);
int ed_lstm_pack(int src_dtype, const void* Whh, void* fwd, void* bwd, int H, hipStream_t s);
int ed_lstm_fwd_fast(void* G, void* Hprev, void* Y, float* Cst, const void* Wfrag, const float* h0,
                     const float* c0, float* hN, float* cN, int B, int Tn, int H, void* ws,
                     hipStream_t s);
int ed_lstm_bwd_fast(void* G, const void* dY, const float* Cst, const float* c0, const void* WTfrag,
                     float* dC, int B, int Tn, int H, void* ws, hipStream_t s);
