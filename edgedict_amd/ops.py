"""Thin tensor-level wrappers over the C ABI (one function per entry point family).

Nothing here computes: each function validates shapes, allocates outputs with the PyTorch
caching allocator and forwards raw pointers to libedgedict_hip.so on the current stream.
"""
import ctypes

import torch

from . import _lib
from ._lib import call, dtype_code, require_cuda


def _ll(x):
    return ctypes.c_longlong(int(x))


def _operand(t):
    """(tensor, ld, k_major) for a 2-D operand that is either row-major or a transposed view."""
    assert t.dim() == 2
    if t.stride(1) == 1:
        return t, t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1]), 1
    if t.stride(0) == 1:
        return t, t.stride(1) if t.shape[1] > 1 else max(t.stride(1), t.shape[0]), 0
    raise ValueError("gemm operand must have a unit stride in one dimension")


def gemm(a, b, out=None, bias=None, bias2=None, accumulate=False, split_k=1,
         out_dtype=None):
    """out[M,N] (+)= a[M,K] @ b[N,K]^T (+ bias).  ``a`` / ``b`` may be transposed *views*
    (``x.t()``): the kernel reads them in place, nothing is materialised."""
    require_cuda(a, b)
    if a.dtype != b.dtype:
        raise TypeError("gemm: operand dtypes differ (%s vs %s)" % (a.dtype, b.dtype))
    M, K = a.shape
    N, K2 = b.shape
    if K != K2:
        raise ValueError("gemm: inner dimensions differ (%d vs %d)" % (K, K2))
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype or a.dtype, device=a.device)
        if accumulate:
            raise ValueError("gemm: accumulate needs an existing out tensor")
    if out.shape != (M, N) or out.stride(1) != 1:
        raise ValueError("gemm: out must be [M,N] with unit column stride")
    a_, lda, akm = _operand(a)
    b_, ldb, bkm = _operand(b)
    for v in (bias, bias2):
        if v is not None and (v.dtype != torch.float32 or v.numel() != N or not v.is_contiguous()):
            raise ValueError("gemm: bias must be contiguous fp32 [N]")
    call("gemm", dtype_code(a.dtype), dtype_code(out.dtype), a_, _ll(lda), akm, b_, _ll(ldb), bkm,
         out, _ll(out.stride(0) if M > 1 else max(out.stride(0), N)), M, N, K, bias, bias2,
         int(bool(accumulate)), int(split_k))
    return out
