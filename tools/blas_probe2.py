"""Vendor GEMM timing on the encoder stack's chunk products (small, latency-critical)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgedict_amd import ops

def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

dt = torch.bfloat16
torch.backends.cuda.preferred_blas_library("hipblaslt")
for rows in (768, 1536):
    dg = torch.randn(rows, 4096, device="cuda").to(dt)
    wt = torch.randn(1024, 4096, device="cuda").to(dt)      # W_ih^T image [I, 4H]: NT product
    x = torch.randn(rows, 1024, device="cuda").to(dt)
    w = torch.randn(4096, 1024, device="cuda").to(dt)
    bias = torch.randn(4096, device="cuda")
    out1 = torch.empty(rows, 1024, device="cuda", dtype=dt)
    out2 = torch.empty(rows, 4096, device="cuda", dtype=dt)
    print("rows %d  dX [r x 1024 x 4096]: vendor %.1f us, own %.1f us | input product [r x 4096 x 1024]: vendor %.1f us, own %.1f us"
          % (rows, t(lambda: torch.mm(dg, wt.t(), out=out1)), t(lambda: ops.gemm(dg, wt, out=out1)),
             t(lambda: torch.addmm(bias.to(dt), x, w.t(), out=out2)), t(lambda: ops.gemm(x, w, bias=bias, out=out2))))
