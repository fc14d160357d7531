#!/usr/bin/env python
"""Background weight-gradient products alone (no recurrence beside them): the joint's dW2 and one encoder
layer's dW_hh / dW_ih, through edgedict_gemm_bg.  EDGEDICT_GEMM_TN256=0|1 and EDGEDICT_BLASLT_BG pick the
kernel (read once per process): run once per setting."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgedict_amd import ops  # noqa: E402

SHAPES = [("joint dW2", 2048, 640, 543526, 8), ("stack dW_hh", 4096, 1024, 8576, 2),
          ("stack dW_ih (2H in)", 4096, 2048, 8576, 2), ("layer0 dW_hh", 4096, 1024, 25664, 2)]
for name, M, N, K, split in SHAPES:
    dy = torch.randn(K, M, device="cuda").bfloat16()
    x = torch.randn(K, N, device="cuda").bfloat16()
    out = torch.zeros(M, N, device="cuda")
    for _ in range(2):
        ops.gemm(dy.t(), x.t(), out=out, accumulate=True, split_k=split, max_wg_per_cu=2)
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        ops.gemm(dy.t(), x.t(), out=out, accumulate=True, split_k=split, max_wg_per_cu=2)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%-22s [%d x %d x %d] split %d: %8.1f us  %6.1f TFLOP/s" % (name, M, N, K, split, dt * 1e6,
                                                                    2.0 * M * N * K / dt / 1e12), flush=True)
    del dy, x, out
