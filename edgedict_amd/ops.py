"""Thin tensor-level wrappers over the C ABI (one function per entry point family).

Nothing here computes: each function validates shapes, allocates outputs with the PyTorch
caching allocator and forwards raw pointers to libedgedict_hip.so on the current stream.
"""
import ctypes
import threading
import time

import torch

from . import _lib
from ._lib import call, dtype_code, require_cuda


def _ll(x):
    return ctypes.c_longlong(int(x))


# ---- optional per-kernel timing with HIP events on the launch stream (used by bench.py) -------
TIMERS = None   # set to {} to enable: tag -> list of (start_event, end_event)


LAST = {}       # last-seen shape facts for bench.py (e.g. rows of the packed joint lattice)
MARKS = None    # set to [] to collect (tag, host seconds, event on the current stream) marks


def mark(tag):
    """Debug aid (tools/host_vs_gpu.py): where is the host, and where is the GPU, at this point?"""
    if MARKS is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        MARKS.append((tag, time.perf_counter(), ev))


HOST = None     # set to {} to accumulate host wall time of selected native calls (bench.py)

# one-shot host callbacks at named points of the step (the trainer hangs the next batch's front-end on one); per host
# thread - one thread per GPU may each drive an engine (nn.DataParallel style), and a hook belongs to its thread's step
_hooks = threading.local()


def set_hook(point, fn):
    """Register ``fn`` to be called once, by this thread, when the step reaches ``point`` (``fire``)."""
    d = getattr(_hooks, "d", None)
    if d is None:
        d = _hooks.d = {}
    d[point] = fn


def pop_hook(point):
    """Remove and return this thread's hook for ``point`` (None if it has fired or was never set)."""
    d = getattr(_hooks, "d", None)
    return d.pop(point, None) if d else None


def fire(point):
    fn = pop_hook(point)
    if fn is not None:
        fn()


class host_timed:
    def __init__(self, tag):
        self.tag = tag

    def __enter__(self):
        if HOST is not None:
            self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        if HOST is not None:
            HOST.setdefault(self.tag, []).append(time.perf_counter() - self.t0)
        return False


def host_summary():
    return {k: (len(v), 1e3 * sum(v) / max(1, len(v))) for k, v in (HOST or {}).items()}


class timed:
    """``with timed("tag"):`` brackets the enclosed launches with events on the current stream
    when ``ops.TIMERS`` is a dict; otherwise it costs nothing."""

    def __init__(self, tag):
        self.tag = tag

    def __enter__(self):
        if TIMERS is not None:
            self.start = torch.cuda.Event(enable_timing=True)
            self.end = torch.cuda.Event(enable_timing=True)
            self.start.record()
        return self

    def __exit__(self, *exc):
        if TIMERS is not None:
            self.end.record()
            TIMERS.setdefault(self.tag, []).append((self.start, self.end))
        return False


def timer_summary():
    """tag -> (count, mean milliseconds); call after a device synchronise."""
    out = {}
    for tag, evs in (TIMERS or {}).items():
        ms = [a.elapsed_time(b) for a, b in evs]
        out[tag] = (len(ms), sum(ms) / max(1, len(ms)))
    return out


def _operand(t):
    """(tensor, ld, k_major) for a 2-D operand that is either row-major or a transposed view."""
    assert t.dim() == 2
    if t.stride(1) == 1:
        return t, t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1]), 1
    if t.stride(0) == 1:
        return t, t.stride(1) if t.shape[1] > 1 else max(t.stride(1), t.shape[0]), 0
    raise ValueError("gemm operand must have a unit stride in one dimension")


def gemm(a, b, out=None, bias=None, bias2=None, accumulate=False, split_k=1,
         out_dtype=None, max_wg_per_cu=0):
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
    args = (dtype_code(a.dtype), dtype_code(out.dtype), a_, _ll(lda), akm, b_, _ll(ldb), bkm,
            out, _ll(out.stride(0) if M > 1 else max(out.stride(0), N)), M, N, K, bias, bias2,
            int(bool(accumulate)), int(split_k))
    if max_wg_per_cu:     # background product on a side stream: capped CU residency, quiet split-K
        part = None
        if out.dtype == torch.float32 and bias is None and bias2 is None:
            nslice = 1
            while nslice < int(split_k) and nslice < 8:
                nslice *= 2
            part = torch.empty(nslice, M, N, dtype=torch.float32, device=out.device)
        call("gemm_bg", *args, int(max_wg_per_cu), part)
    else:
        call("gemm", *args)
    return out


def cast(x, dtype):
    """Contiguous dtype conversion (fp32 <-> bf16) on the device."""
    require_cuda(x)
    if x.dtype == dtype:
        return x
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    call("cast", dtype_code(x.dtype), x, dtype_code(dtype), out, _ll(x.numel()))
    return out


def transpose(x, dtype=None):
    """out[c, r] = x[r, c] (optionally converting dtype). x must be a contiguous 2-D tensor."""
    require_cuda(x)
    assert x.dim() == 2 and x.is_contiguous()
    dtype = dtype or x.dtype
    out = torch.empty(x.shape[1], x.shape[0], dtype=dtype, device=x.device)
    call("transpose", dtype_code(x.dtype), x, dtype_code(dtype), out, x.shape[0], x.shape[1])
    return out


def colsum(x2d, out=None):
    """fp32 column sums of a row-major [M,N] matrix (atomically accumulated into ``out``)."""
    require_cuda(x2d)
    M, N = x2d.shape
    assert x2d.stride(1) == 1
    if out is None:
        out = torch.zeros(N, dtype=torch.float32, device=x2d.device)
    call("colsum", dtype_code(x2d.dtype), x2d, _ll(x2d.stride(0) if M > 1 else N), out, _ll(M), N)
    return out


def layernorm_fwd(x, res, gamma, beta, reduce=1, eps=1e-5):
    """y = pairmean_reduce(LN(x + res)); x,res [B,T,D] in the compute dtype."""
    require_cuda(x, gamma, beta)
    B, T, D = x.shape
    assert x.is_contiguous() and (res is None or (res.is_contiguous() and res.shape == x.shape))
    Tout = (T + reduce - 1) // reduce
    y = torch.empty(B, Tout, D, dtype=x.dtype, device=x.device)
    mean = torch.empty(B * T, dtype=torch.float32, device=x.device)
    rstd = torch.empty(B * T, dtype=torch.float32, device=x.device)
    call("layernorm_fwd", dtype_code(x.dtype), x, res, gamma, beta, y, mean, rstd, B, T, D,
         int(reduce), float(eps))
    return y, mean, rstd


def layernorm_bwd(dout, x, res, gamma, mean, rstd, reduce=1, dgamma=None, dbeta=None):
    require_cuda(dout, x)
    B, T, D = x.shape
    dout = dout.contiguous()
    ds = torch.empty_like(x)
    if dgamma is None:
        dgamma = torch.zeros(D, dtype=torch.float32, device=x.device)
    if dbeta is None:
        dbeta = torch.zeros(D, dtype=torch.float32, device=x.device)
    call("layernorm_bwd", dtype_code(x.dtype), dout, x, res, gamma, mean, rstd, ds, dgamma, dbeta,
         B, T, D, int(reduce))
    return ds, dgamma, dbeta


def lstm_pack_weights(Whh, want_fwd=True, want_bwd=True):
    """Fragment-order bf16 images of W_hh [4H,H] for the fast recurrence kernels."""
    require_cuda(Whh)
    H4, H = Whh.shape
    assert H4 == 4 * H and Whh.is_contiguous()
    dev = Whh.device
    fwd = torch.empty(H4 * H, dtype=torch.bfloat16, device=dev) if want_fwd else None
    bwd = torch.empty(H4 * H, dtype=torch.bfloat16, device=dev) if want_bwd else None
    call("lstm_pack_weights", dtype_code(Whh.dtype), Whh, fwd, bwd, H)
    return fwd, bwd


def _lstm_ws(dtype, B, H, device):
    n = _lib.load().edgedict_lstm_workspace_bytes(dtype_code(dtype), B, H)
    return torch.empty(n, dtype=torch.uint8, device=device) if n else None


def lstm_forward(G, Whh, h0=None, c0=None, Whh_packed=None):
    """Run the recurrence over G[B,T,4H] (input pre-activations, overwritten with the gates).
    ``Whh_packed`` (from lstm_pack_weights) selects the bf16 fragment-order fast path.
    Returns (Y, Hprev, Cst, hN, cN)."""
    require_cuda(G)
    B, T, H4 = G.shape
    H = H4 // 4
    assert G.is_contiguous()
    assert Whh is None or (Whh.is_contiguous() and Whh.shape == (H4, H))
    dev = G.device
    Hprev = torch.empty(B, T, H, dtype=G.dtype, device=dev)
    Y = torch.empty(B, T, H, dtype=G.dtype, device=dev)
    Cst = torch.empty(B, T, H, dtype=torch.float32, device=dev)
    hN = torch.empty(B, H, dtype=torch.float32, device=dev)
    cN = torch.empty(B, H, dtype=torch.float32, device=dev)
    for s in (h0, c0):
        if s is not None:
            assert s.dtype == torch.float32 and s.is_contiguous() and s.shape == (B, H)
    ws = _lstm_ws(G.dtype, B, H, dev) if Whh_packed is not None else None
    call("lstm_forward", dtype_code(G.dtype), G, Hprev, Y, Cst, Whh, Whh_packed, h0, c0, hN, cN,
         B, T, H, ws)
    return Y, Hprev, Cst, hN, cN


def lstm_backward(G, dY, Cst, c0, WhhT, WhhT_packed=None):
    """BPTT sweep: G (saved gates) is overwritten with dL/d(pre-activation) for every step."""
    B, T, H4 = G.shape
    H = H4 // 4
    assert WhhT is None or (WhhT.shape == (H, H4) and WhhT.is_contiguous())
    if dY is not None:
        assert dY.is_contiguous() and dY.dtype == G.dtype and dY.shape == (B, T, H)
    dC = torch.empty(B, H, dtype=torch.float32, device=G.device)
    ws = _lstm_ws(G.dtype, B, H, G.device) if WhhT_packed is not None else None
    call("lstm_backward", dtype_code(G.dtype), G, dY, Cst, c0, WhhT, WhhT_packed, dC, B, T, H, ws)
    return G


def conv_out_frames(Tin, k, s):
    return int(_lib.load().edgedict_conv_out_frames(int(Tin), int(k), int(s)))


def conv_im2col(x, k, s, out_dtype):
    """x [B,Tin,C] channels-last -> cols [B,Tout,C*k] (column c*k + j reads frame t*s - (k-1) + j)."""
    require_cuda(x)
    B, Tin, C = x.shape
    assert x.is_contiguous()
    Tout = conv_out_frames(Tin, k, s)
    if Tout <= 0:
        raise ValueError("conv: %d input frames are too few for kernel %d / stride %d" % (Tin, k, s))
    cols = torch.empty(B, Tout, C * k, dtype=out_dtype, device=x.device)
    call("conv_im2col", dtype_code(x.dtype), dtype_code(out_dtype), x, cols, B, Tin, C, int(k), int(s))
    return cols


def conv_col2im(dcols, Tin, C, k, s):
    B, Tout, CK = dcols.shape
    assert dcols.is_contiguous() and CK == C * k
    dx = torch.empty(B, Tin, C, dtype=dcols.dtype, device=dcols.device)
    call("conv_col2im", dtype_code(dcols.dtype), dcols, dx, B, int(Tin), int(C), int(k), int(s))
    return dx


def gelu_groupnorm_fwd(y, gamma, beta, eps=1e-5):
    require_cuda(y)
    B, T, C = y.shape
    assert y.is_contiguous()
    out = torch.empty_like(y)
    mean = torch.empty(B, dtype=torch.float32, device=y.device)
    rstd = torch.empty(B, dtype=torch.float32, device=y.device)
    call("gelu_groupnorm_fwd", dtype_code(y.dtype), y, gamma, beta, out, mean, rstd, B, T, C, float(eps))
    return out, mean, rstd


def gelu_groupnorm_bwd(y, dout, gamma, mean, rstd):
    B, T, C = y.shape
    dout = dout.contiguous()
    dy = torch.empty_like(y)
    dgamma = torch.empty(C, dtype=torch.float32, device=y.device)
    dbeta = torch.empty(C, dtype=torch.float32, device=y.device)
    ws = torch.empty(_lib.load().edgedict_gelu_groupnorm_bwd_workspace_bytes(B, T, C), dtype=torch.uint8,
                     device=y.device)
    call("gelu_groupnorm_bwd", dtype_code(y.dtype), y, dout, gamma, mean, rstd, dy, dgamma, dbeta, ws,
         B, T, C)
    return dy, dgamma, dbeta


def gru_forward(G, Whh, b_hh, h0=None):
    """GRU recurrence over G[B,T,3H] (= x W_ih^T + b_ih, overwritten with r,z,n).
    Returns (Y, Hprev, HN, hN)."""
    require_cuda(G)
    B, T, H3 = G.shape
    H = H3 // 3
    assert G.is_contiguous() and Whh.is_contiguous() and Whh.shape == (H3, H) and Whh.dtype == G.dtype
    assert b_hh.dtype == torch.float32 and b_hh.shape == (H3,)
    dev = G.device
    Hprev = torch.empty(B, T, H, dtype=G.dtype, device=dev)
    Y = torch.empty(B, T, H, dtype=G.dtype, device=dev)
    HN = torch.empty(B, T, H, dtype=G.dtype, device=dev)
    hN = torch.empty(B, H, dtype=torch.float32, device=dev)
    if h0 is not None:
        assert h0.dtype == torch.float32 and h0.is_contiguous() and h0.shape == (B, H)
    call("gru_forward", dtype_code(G.dtype), G, Hprev, Y, HN, Whh, b_hh.contiguous(), h0, hN, B, T, H)
    return Y, Hprev, HN, hN


def gru_backward(G, dY, Hprev, HN, WhhT):
    """BPTT sweep: G (saved r,z,n) becomes the input-side pre-activation gradient; returns DH, the
    hidden-side one."""
    B, T, H3 = G.shape
    H = H3 // 3
    assert WhhT.shape == (H, H3) and WhhT.is_contiguous() and WhhT.dtype == G.dtype
    if dY is not None:
        assert dY.is_contiguous() and dY.dtype == G.dtype and dY.shape == (B, T, H)
    DH = torch.empty_like(G)
    dh = torch.empty(B, H, dtype=torch.float32, device=G.device)
    call("gru_backward", dtype_code(G.dtype), G, DH, dY, Hprev, HN, WhhT, dh, B, T, H)
    return DH


def embedding_fwd(tokens, weight, out_dtype, prepend_bos, bos):
    """tokens int32 [B,U] -> [B, U(+1), E] in out_dtype; weight may be the fp32 master."""
    require_cuda(weight)
    B, U = tokens.shape
    V, E = weight.shape
    Uout = U + (1 if prepend_bos else 0)
    out = torch.empty(B, Uout, E, dtype=out_dtype, device=weight.device)
    tok = tokens if U > 0 else None
    call("embedding_fwd", dtype_code(out_dtype), dtype_code(weight.dtype), tok,
         int(tokens.stride(0)) if U > 0 else 0, weight, out, B, Uout, E, V, int(prepend_bos),
         int(bos))
    return out


def embedding_bwd(tokens, dout, V, prepend_bos, bos, pad):
    B, Uout, E = dout.shape
    U = tokens.shape[1]
    demb = torch.zeros(V, E, dtype=torch.float32, device=dout.device)
    dout = dout.contiguous()
    tok = tokens if U > 0 else None
    call("embedding_bwd", dtype_code(dout.dtype), tok, int(tokens.stride(0)) if U > 0 else 0,
         dout, demb, B, Uout, E, V, int(prepend_bos), int(bos), int(pad))
    return demb


def joint_hidden_fwd(E1, D1):
    B, T, J = E1.shape
    U1 = D1.shape[1]
    hid = torch.empty(B, T, U1, J, dtype=E1.dtype, device=E1.device)
    call("joint_hidden_fwd", dtype_code(E1.dtype), E1, D1, hid, B, T, U1, J)
    return hid


def joint_hidden_bwd(dhid, hid):
    B, T, U1, J = hid.shape
    dE1 = torch.empty(B, T, J, dtype=torch.float32, device=hid.device)
    dD1 = torch.empty(B, U1, J, dtype=torch.float32, device=hid.device)
    call("joint_hidden_bwd", dtype_code(hid.dtype), dhid, hid, dE1, dD1, B, T, U1, J)
    return dE1, dD1


def pick_split_k(M, N, K, bk=64):
    """Split-K factor for weight-gradient GEMMs (small M*N, huge K): aim at >= ~512 workgroups."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    want = max(1, 768 // max(tiles, 1))
    return int(max(1, min(want, K // (4 * bk) if K >= 8 * bk else 1)))


def dropout(x, p, seed, out=None):
    """Counter-based dropout (training mode): ``x * keep / (1 - p)``; the same (p, seed) applied to
    a gradient is the backward pass."""
    require_cuda(x)
    x = x.contiguous()
    if out is None:
        out = torch.empty_like(x)
    call("dropout", dtype_code(x.dtype), x, out, _ll(x.numel()), float(p),
         ctypes.c_uint(int(seed) & 0xFFFFFFFF))
    return out


def spec_mask_(x, t_intervals, f_intervals):
    """In-place SpecAugment zero-fill on a resident fp32 feature batch [B,T,F].
    ``t_intervals`` / ``f_intervals``: int32 device tensors [B, n, 2] of half-open intervals over
    the frame index / the (stacked) feature index, or None."""
    require_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 3
    B, T, F_ = x.shape
    n_t = 0 if t_intervals is None else t_intervals.shape[1]
    n_f = 0 if f_intervals is None else f_intervals.shape[1]
    call("spec_mask", x, B, T, F_, t_intervals, n_t, f_intervals, n_f)
    return x
