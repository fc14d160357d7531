"""How fast are the vendor GEMM libraries (through torch.mm / addmm, i.e. hipBLASLt / rocBLAS) on the
joint's three big products?  Decides whether the plain products should call the library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgedict_amd import ops

def t(fn, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

M, J, V = 543526, 640, 2048
dt = torch.bfloat16
hid = torch.randn(M, J, device="cuda").to(dt)
w2 = torch.randn(V, J, device="cuda").to(dt)
b2 = torch.randn(V, device="cuda").to(dt)
b2f = b2.float()
g = torch.randn(M, V, device="cuda").to(dt)
out = torch.empty(M, V, device="cuda", dtype=dt)
for pref in ("default", "hipblaslt", "hipblas"):
    try:
        if pref != "default": torch.backends.cuda.preferred_blas_library(pref)
    except Exception as ex:
        print(pref, "unavailable", ex); continue
    ms = t(lambda: torch.addmm(b2, hid, w2.t(), out=out))
    print("%-10s logits addmm [%d x %d x %d]: %.3f ms  %.0f TF/s" % (pref, M, V, J, ms, 2.0 * M * V * J / ms / 1e9))
    ms = t(lambda: torch.mm(g, w2))
    print("%-10s dhid   mm    [%d x %d x %d]: %.3f ms  %.0f TF/s" % (pref, M, J, V, ms, 2.0 * M * V * J / ms / 1e9))
    ms = t(lambda: torch.mm(g.t(), hid), iters=3)
    print("%-10s dW2    mm TN [%d x %d x %d]: %.3f ms  %.0f TF/s" % (pref, V, J, M, ms, 2.0 * M * V * J / ms / 1e9))
    x = torch.randn(25664, 1024, device="cuda").to(dt); gg = torch.randn(25664, 4096, device="cuda").to(dt)
    ms = t(lambda: torch.mm(gg.t(), x))
    print("%-10s enc dW TN    [4096 x 1024 x 25664]: %.3f ms  %.0f TF/s" % (pref, ms, 2.0 * 4096 * 1024 * 25664 / ms / 1e9))
ms = t(lambda: ops.gemm(hid, w2, bias=b2f, out=out))
print("ours       logits NT: %.3f ms  %.0f TF/s" % (ms, 2.0 * M * V * J / ms / 1e9))
