#!/usr/bin/env python
"""Would pipelining the joint's backward over two utterance halves pay?  The HBM-bound gradient kernel of one
half (rnnt_grad: reads 2.2 GB of logits, writes 2.2 GB) beside the fetch/MFMA-bound dhid product of the other
(gemm_nt256 [M x 640 x 2048]).  Stand-ins at half size on two streams: a device copy of M/2 x 2048 bf16 and the
real product on M/2 rows; prints each alone, both serially, both concurrently."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgedict_amd import ops, side  # noqa: E402

M, V, J = 543526 // 2, 2048, 640
dev = torch.device("cuda", 0)
side.stream(dev)
s2 = side.stream(dev)
dl = torch.randn(M, V, device=dev).bfloat16()
src = torch.randn(M, V, device=dev).bfloat16()
dst = torch.empty_like(src)
w2t = (0.05 * torch.randn(J, V, device=dev)).bfloat16()
out = torch.empty(M, J, device=dev, dtype=torch.bfloat16)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def gemm():
    ops.gemm(dl, w2t, out=out)


def copy():
    dst.copy_(src)


def both():
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        dst.copy_(src)
    ops.gemm(dl, w2t, out=out)
    torch.cuda.current_stream().wait_stream(s2)


a, b = timeit(gemm), timeit(copy)
c = timeit(lambda: (gemm(), copy()))
d = timeit(both)
print("dhid product (half) %.3f ms, 2 x 1.1 GB copy %.3f ms, serial %.3f ms, concurrent %.3f ms" % (a, b, c, d))
