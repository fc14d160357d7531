"""Host side of the layer-pipelined encoder stack (csrc/encoder_stack.hip).

``EncoderStackFn`` is the autograd node that stands in for the reference's
``Encoder.forward`` body up to (not including) the output projection — input LayerNorm and the
``ResLayerNormLSTM`` loop, rnnt/models.py:55-75,124,131-134 — when the compute dtype is bf16.
It allocates every buffer with the caching allocator, fills the C descriptors
(``edgedict_stack_desc_t`` / ``edgedict_stack_layer_t`` in include/edgedict_hip.h) and makes ONE
native call per direction; the stream scheduling lives in the library.
"""
import ctypes
import os
import weakref

import torch

from . import _lib, config, ops
from ._lib import check, dtype_code, ptr, stream_ptr

F32 = torch.float32
BF16 = torch.bfloat16
_vp, _fp = ctypes.c_void_p, ctypes.c_void_p


class StackLayer(ctypes.Structure):
    _fields_ = [("T", ctypes.c_int), ("I", ctypes.c_int), ("reduce", ctypes.c_int),
                ("residual", ctypes.c_int),
                ("wih_p", _vp), ("wih_t", _vp), ("bias_p", _fp), ("whh_f", _vp), ("whh_b", _vp),
                ("ln_gamma", _fp), ("ln_beta", _fp),
                ("X", _vp), ("G", _vp), ("Yx", _vp), ("Cx", _fp), ("mean", _fp), ("rstd", _fp),
                ("dZ", _vp), ("dX", _vp), ("dW_ih", _fp), ("dW_hh", _fp), ("db", _fp), ("db_hh", _fp),
                ("dgamma", _fp), ("dbeta", _fp), ("whh_s", _vp),
                ("drop_p", ctypes.c_float), ("drop_seed", ctypes.c_uint)]


class StackDesc(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int), ("H", ctypes.c_int), ("L", ctypes.c_int),
                ("chunk", ctypes.c_int), ("lag", ctypes.c_int), ("split_k", ctypes.c_int),
                ("flags", ctypes.c_int), ("eps", ctypes.c_float),
                ("layers", ctypes.POINTER(StackLayer)),
                ("x", _vp), ("x_dtype", ctypes.c_int), ("T0", ctypes.c_int), ("I0", ctypes.c_int),
                ("in_gamma", _fp), ("in_beta", _fp), ("in_mean", _fp), ("in_rstd", _fp),
                ("h0", _fp), ("c0", _fp), ("out", _vp), ("dout", _vp),
                ("d_in_gamma", _fp), ("d_in_beta", _fp), ("ws", _vp), ("ws_bytes", ctypes.c_size_t),
                ("grads_final", _vp), ("grads_final_user", _vp)]


GRADS_FINAL_CB = ctypes.CFUNCTYPE(None, ctypes.c_int, ctypes.c_void_p)


SERIAL = 1
DW_AT_END = 2
ACCUM_GRADS = 8
INFERENCE = 64     # no backward pass will follow (torch.no_grad): no dG images / split-K partials in the workspace

# schedule knobs (env overrides are for tuning runs; the defaults are what bench.py measures)
# output-rate frames per chunk (= steps per launch of the launch-persistent kernels).  E6D2 training step, round 3:
# 8 / 10 / 12 / 16 / 20 / 24 / 32 frames = 22.8 / 22.2 / 21.2 / 21.0 / 21.3 / 21.4 / 22.1 ms
CHUNK = int(os.environ.get("EDGEDICT_STACK_CHUNK", "16"))
LAG = int(os.environ.get("EDGEDICT_STACK_LAG", "0"))
SPLIT_K = int(os.environ.get("EDGEDICT_STACK_SPLITK", "0"))
FLAGS = int(os.environ.get("EDGEDICT_STACK_FLAGS", "0"))
# pack the split-K image of W_hh (8 MB per layer) for the weights-stationary BPTT kernel whenever the geometry
# allows it; the kernel itself is chosen per call by the library (EDGEDICT_STACK_BWD_SK, read there)
BWD_SK = os.environ.get("EDGEDICT_STACK_BWD_SK_PACK", "1") != "0"


def _p(t):
    return None if t is None else t.data_ptr()


def side_stream(device):
    from . import side
    return side.stream(device)


def supported(cd, H, I0, L, reductions):
    return (cd == BF16 and H % 32 == 0 and 32 <= H <= 2048 and I0 % 8 == 0 and 8 <= I0 <= 1024
            and 1 <= L <= 8 and all(r in (1, 2) for r in reductions) and config.USE_ENCODER_STACK)


class _PackedLayer:
    """bf16 weight images of one LSTM layer, rebuilt when the fp32 masters change: the forward pass's images (wih_p,
    bias_p, whh_f) under ``key``, the ones only the backward pass reads (wih_t, whh_b, whh_s) under ``bwd_key``."""
    __slots__ = ("key", "bwd_key", "ref", "wih_p", "wih_t", "bias_p", "whh_f", "whh_b", "whh_s")

    def __init__(self, owner):
        self.key = self.bwd_key = None
        self.ref = weakref.ref(owner)


_PACKED = {}


def _sources(*ts):
    srcs = [t.detach().contiguous() for t in ts]
    for t in srcs:
        if t.dtype != F32:
            raise TypeError("encoder stack: master weights must be fp32")
    return srcs


def packed_weights(w_ih, w_hh, b_ih, b_hh, backward_images=True):
    """The layer's images, (re)built on the current stream when the masters changed (a NEW entry then: descriptors that
    hold pointers into the old one keep reading consistent images).  ``backward_images=False`` leaves wih_t / whh_b /
    whh_s allocated but not built - their pointers can go into a descriptor, and ``fill_backward_images`` builds them
    before the backward pass reads them (``_Plan.backward``; ``prepack`` does it behind the forward images)."""
    ent = _PACKED.get(id(w_hh))
    key = (w_ih.data_ptr(), w_hh.data_ptr(), w_ih._version, w_hh._version, b_ih._version,
           b_hh._version, config.param_epoch(), BWD_SK)
    if ent is None or ent.ref() is not w_hh or ent.key != key:     # (ids are recycled: the entry must be this tensor's)
        for k in [k for k, v in _PACKED.items() if v.ref() is None]:
            del _PACKED[k]                       # images of parameters that no longer exist
        ent = _PACKED[id(w_hh)] = _PackedLayer(w_hh)
        H4, I = w_ih.shape
        H = H4 // 4
        dev = w_ih.device
        srcs = _sources(w_ih, w_hh, b_ih, b_hh)
        ent.wih_p = torch.empty(H4, I, dtype=BF16, device=dev)
        ent.wih_t = torch.empty(I, H4, dtype=BF16, device=dev)
        ent.bias_p = torch.empty(H4, dtype=F32, device=dev)
        ent.whh_f = torch.empty(H4 * H, dtype=BF16, device=dev)
        ent.whh_b = torch.empty(H4 * H, dtype=BF16, device=dev)
        # split-K image of the weights-stationary BPTT kernel (csrc/stack_kernels.hip stack_bwd_sk_kernel)
        ent.whh_s = torch.empty(H4 * H, dtype=BF16, device=dev) if (BWD_SK and H % 64 == 0 and 64 <= H <= 1024) else None
        check(_lib.load().edgedict_stack_pack_weights(ptr(srcs[0]), ptr(srcs[1]), ptr(srcs[2]), ptr(srcs[3]),
                                                      H, I, ptr(ent.wih_p), None, ptr(ent.bias_p), ptr(ent.whh_f),
                                                      None, stream_ptr()), "stack_pack_weights")
        ent.key = key
    if backward_images:
        fill_backward_images(ent, w_ih, w_hh)
    return ent


def fill_backward_images(ent, w_ih, w_hh):
    """Build wih_t / whh_b / whh_s of ``ent`` on the current stream unless they are there already."""
    if ent.bwd_key == ent.key:
        return
    H4, I = w_ih.shape
    H = H4 // 4
    srcs = _sources(w_ih, w_hh)
    lib = _lib.load()
    check(lib.edgedict_stack_pack_weights(ptr(srcs[0]), ptr(srcs[1]), None, None, H, I, None, ptr(ent.wih_t), None,
                                          None, ptr(ent.whh_b), stream_ptr()), "stack_pack_weights (backward images)")
    if ent.whh_s is not None:
        check(lib.edgedict_stack_pack_sk(ptr(srcs[1]), H, ptr(ent.whh_s), stream_ptr()), "stack_pack_sk")
    ent.bwd_key = ent.key


def clear_cache():
    _PACKED.clear()


def check_wsr_error():
    """Raise if a launch on this device ran into one of its bounded in-kernel waits - a launch-persistent
    recurrence launch whose workgroups could not all become resident or whose peer never arrived (7xx forward,
    9xx split-K BPTT), a side stream's counter wait (8xx), or a step kernel whose chunk product never signalled
    (5xx / 6xx, csrc/stack_kernels.hip soft_wait): the results of that call are garbage.  The code is a pinned
    host word the device writes; reading it costs nothing and clears it.  Call after a synchronize for an
    up-to-date answer."""
    code = _lib.load().edgedict_stack_wsr_error()
    if 500 <= code < 700:
        raise RuntimeError("edgedict_amd: a step kernel of the encoder stack gave up waiting for a chunk product "
                           "of the side stream (code %d: %s pass, launch slot %d); the results of that call are "
                           "garbage. EDGEDICT_STACK_SOFT_WAIT=0 orders the streams with events instead"
                           % (code, "forward" if code < 600 else "backward", code % 100))
    if code:
        raise RuntimeError("edgedict_amd: a launch-persistent encoder launch gave up waiting for its peers (code %d); "
                           "the results of that call are garbage. EDGEDICT_STACK_LPW=0 / EDGEDICT_STACK_BWD_SK=0 "
                           "select the launch-per-step kernels (e.g. when two processes share the device)" % code)


def last_mode(backward):
    """(kind, steps_per_launch) of the recurrence kernel the most recent forward / backward call of the stack ran on
    this device: kind 0 = one launch per time step, 1 = launch-persistent forward, 2 = split-K weights-stationary BPTT
    (edgedict_stack_last_mode).  Tests and bench.py assert the benched path with it: a silent fallback to the
    launch-per-step kernels would still be numerically right."""
    kind, steps = ctypes.c_int(-1), ctypes.c_int(-1)
    check(_lib.load().edgedict_stack_last_mode(1 if backward else 0, ctypes.byref(kind), ctypes.byref(steps)),
          "stack_last_mode")
    return kind.value, steps.value


_PREPACK_EVENT = [None, None]     # images of the forward pass / of the backward pass rebuilt on another stream


def prepack(encoder, stream):
    """Rebuild the weight images of ``encoder`` (an ``Encoder`` whose ``lstm`` is a
    ``ResLayerNormLSTM``) on ``stream`` - the trainer calls this on the auxiliary stream right after
    the optimiser step.  The next forward pass cannot start before its images exist, so those come first
    (~60 us; the first plan built afterwards waits for their event) and the backward pass's images behind
    them (its first call waits for the second event, long complete by then).  Round 6: 0.39 -> ~0.06 ms
    between the optimiser step and the encoder (profiles/r6_pack.txt)."""
    lstm = encoder.lstm
    if not hasattr(lstm, "lstms"):
        return
    with torch.cuda.stream(stream):
        # what the forward pass reads first (the wait in front of the encoder is this part's: ~60 of the ~150 us), then
        # what only the backward pass reads
        ents = [(packed_weights(*m.layer(0), backward_images=False), m.layer(0)) for m in lstm.lstms]
        _PREPACK_EVENT[0] = stream.record_event()
        for ent, lay in ents:
            fill_backward_images(ent, lay[0], lay[1])
        _PREPACK_EVENT[1] = stream.record_event()


class _Plan:
    """Buffers + descriptors of one forward/backward pair."""

    def __init__(self, x, in_norm, layers, reductions, h0, c0, flags, drop=None):
        self.keep = []          # tensors referenced by raw pointer from the descriptors
        # drop: None, or (p, [seed of layer 0, ...]): nn.Dropout behind every layer's LayerNorm (+ TimeReduction),
        # rnnt/models.py:47-53,70, applied inside the norm role / the LayerNorm backward (training mode only)
        if _PREPACK_EVENT[0] is not None:      # images were rebuilt on another stream (prepack)
            torch.cuda.current_stream(x.device).wait_event(_PREPACK_EVENT[0])
            _PREPACK_EVENT[0] = None
        B, T0, I0 = x.shape
        dev = x.device
        L = len(layers)
        H = layers[0][1].shape[1]
        self.B, self.H, self.L, self.T0, self.I0 = B, H, L, T0, I0
        self.x = x
        self.in_mean = torch.empty(B * T0, dtype=F32, device=dev)
        self.in_rstd = torch.empty(B * T0, dtype=F32, device=dev)
        self.larr = (StackLayer * L)()
        self.layer_bufs = []
        self.packs = []         # (weight images, masters) per layer: the backward pass's images may still have to be built
        T, I = T0, I0
        for l, (w_ih, w_hh, b_ih, b_hh, ln_w, ln_b) in enumerate(layers):
            pk = packed_weights(w_ih, w_hh, b_ih, b_hh, backward_images=False)
            self.packs.append((pk, w_ih, w_hh))
            bufs = dict(
                X=torch.empty(T, B, I, dtype=BF16, device=dev),
                G=torch.empty(T, B, 4 * H, dtype=BF16, device=dev),
                Yx=torch.empty(T + 1, B, H, dtype=BF16, device=dev),
                Cx=torch.empty(T + 1, B, H, dtype=F32, device=dev),
                mean=torch.empty(T, B, dtype=F32, device=dev),
                rstd=torch.empty(T, B, dtype=F32, device=dev))
            self.layer_bufs.append(bufs)
            y = self.larr[l]
            y.T, y.I, y.reduce, y.residual = T, I, reductions[l], int(l != 0)
            y.wih_p, y.wih_t, y.bias_p, y.whh_f, y.whh_b = map(
                _p, (pk.wih_p, pk.wih_t, pk.bias_p, pk.whh_f, pk.whh_b))
            y.whh_s = _p(pk.whh_s)
            y.drop_p, y.drop_seed = (float(drop[0]), int(drop[1][l]) & 0xFFFFFFFF) if drop else (0.0, 0)
            g, b = ln_w.detach(), ln_b.detach()
            y.ln_gamma, y.ln_beta = _p(g), _p(b)
            self.keep += [pk.wih_p, pk.wih_t, pk.bias_p, pk.whh_f, pk.whh_b, pk.whh_s, g, b]
            for k, v in bufs.items():
                setattr(y, k, _p(v))
            T = (T + reductions[l] - 1) // reductions[l]
            I = H
        self.T_out = T
        self.out = torch.empty(B, T, H, dtype=BF16, device=dev)
        d = self.desc = StackDesc()
        d.B, d.H, d.L = B, H, L
        d.chunk, d.lag, d.split_k, d.flags, d.eps = CHUNK, LAG, SPLIT_K, flags, 1e-5
        d.layers = ctypes.cast(self.larr, ctypes.POINTER(StackLayer))
        d.x, d.x_dtype, d.T0, d.I0 = _p(x), dtype_code(x.dtype), T0, I0
        ig, ib = in_norm[0].detach(), in_norm[1].detach()
        self.keep += [ig, ib, h0, c0]
        d.in_gamma, d.in_beta = _p(ig), _p(ib)
        d.in_mean, d.in_rstd = _p(self.in_mean), _p(self.in_rstd)
        d.h0, d.c0 = _p(h0), _p(c0)
        d.out = _p(self.out)
        lib = _lib.load()
        d.ws, d.ws_bytes = None, 0
        n = lib.edgedict_stack_workspace_bytes(ctypes.byref(d))
        self.ws = torch.empty(n, dtype=torch.uint8, device=dev)
        d.ws, d.ws_bytes = _p(self.ws), n

    def forward(self):
        lib = _lib.load()
        check_wsr_error()      # a PREVIOUS weights-stationary launch that gave up (host word, no sync)
        ops.mark("stack_fwd:enter")
        with ops.host_timed("stack_forward_call"):
            check(lib.edgedict_stack_forward(ctypes.byref(self.desc), stream_ptr()), "stack_forward")
        ops.mark("stack_fwd:exit")

    def final_states(self):
        hN = torch.stack([b["Yx"][-1] for b in self.layer_bufs], 0).float()
        cN = torch.stack([b["Cx"][-1] for b in self.layer_bufs], 0).clone()
        return hN, cN

    def backward(self, dout, params, in_norm):
        """Returns (dig, dib, per-layer grads) or None when every gradient was accumulated straight
        into the parameters' existing fp32 .grad buffers (flat-buffer training: no AccumulateGrad
        adds, no temporaries)."""
        dev = dout.device
        B, H = self.B, self.H
        grads = []
        d = self.desc
        if _PREPACK_EVENT[1] is not None:      # the backward pass's weight images were rebuilt on another stream (prepack)
            torch.cuda.current_stream(dev).wait_event(_PREPACK_EVENT[1])
            _PREPACK_EVENT[1] = None
        for pk, w_ih, w_hh in self.packs:      # ... or not at all yet (no prepack: first step, plain autograd use)
            fill_backward_images(pk, w_ih, w_hh)
        everyone = list(params) + list(in_norm)
        direct = config.DEFER_WEIGHT_GRADS and all(
            p.grad is not None and p.grad.dtype == F32 and p.grad.is_contiguous() for p in everyone)
        zeros = None if direct else torch.zeros(2 * H * self.L + 2 * self.I0, dtype=F32, device=dev)
        for l in range(self.L):
            y = self.larr[l]
            T, I = y.T, y.I
            w_ih, w_hh, b_ih, b_hh, ln_w, ln_b = params[6 * l:6 * l + 6]
            dZ = torch.empty(T, B, H, dtype=BF16, device=dev)
            gb = dict(dZ=dZ, dX=dZ if y.residual else (torch.empty(T, B, I, dtype=BF16, device=dev)
                                                       if l > 0 else None))
            if direct:
                gb.update(dW_ih=w_ih.grad, dW_hh=w_hh.grad, db=b_ih.grad, db_hh=b_hh.grad,
                          dgamma=ln_w.grad, dbeta=ln_b.grad)
            else:
                gb.update(dW_ih=torch.empty(4 * H, I, dtype=F32, device=dev),
                          dW_hh=torch.empty(4 * H, H, dtype=F32, device=dev),
                          db=torch.empty(4 * H, dtype=F32, device=dev), db_hh=None,
                          dgamma=zeros[2 * H * l:2 * H * l + H],
                          dbeta=zeros[2 * H * l + H:2 * H * (l + 1)])
            for k, v in gb.items():
                setattr(y, k, _p(v))
            grads.append(gb)
        if direct:
            dig, dib = in_norm[0].grad, in_norm[1].grad
            d.flags |= ACCUM_GRADS
        else:
            dig = zeros[2 * H * self.L:2 * H * self.L + self.I0]
            dib = zeros[2 * H * self.L + self.I0:]
            d.flags &= ~ACCUM_GRADS
        dout = dout.contiguous()
        d.dout, d.d_in_gamma, d.d_in_beta = _p(dout), _p(dig), _p(dib)
        # data parallelism: tell the gradient exchange when a layer's weight gradients are final on the
        # auxiliary stream, so its bucket leaves while the layers below are still in their BPTT
        from . import dp
        cb = None
        if direct and dp.READY_HOOK is not None:
            hook, aux = dp.READY_HOOK, side_stream(dev)

            errors = []

            def _on_final(layer, _user, params=params, hook=hook, aux=aux):
                try:        # an exception must not unwind through the C frames of the scheduler
                    hook(params[6 * layer:6 * layer + 4], aux)
                except BaseException as exc:   # noqa: BLE001 - re-raised below
                    errors.append(exc)
            cb = GRADS_FINAL_CB(_on_final)
            d.grads_final = ctypes.cast(cb, ctypes.c_void_p).value
        else:
            d.grads_final = None
        d.grads_final_user = None
        lib = _lib.load()
        ops.mark("stack_bwd:enter")
        with ops.host_timed("stack_backward_call"):
            check(lib.edgedict_stack_backward(ctypes.byref(d), stream_ptr()), "stack_backward")
        d.grads_final = None
        if cb is not None and errors:
            raise errors[0]
        ops.mark("stack_bwd:exit")
        return None if direct else (dig, dib, grads)


class EncoderStackFn(torch.autograd.Function):
    """(xs [B,T0,I0], in_gamma, in_beta, h0, c0, reductions, flags, drop, 6 tensors per layer...)
    -> (out [B,T',H] bf16, hN [L,B,H] f32, cN [L,B,H] f32);  drop = None or (p, per-layer seeds)"""

    @staticmethod
    def forward(ctx, xs, in_g, in_b, h0, c0, reductions, flags, drop, *params):
        assert len(params) % 6 == 0
        layers = [params[i:i + 6] for i in range(0, len(params), 6)]
        x = xs.contiguous()
        if x.dtype not in (F32, BF16):
            x = x.float()
        needs_bwd = any(ctx.needs_input_grad)        # (grad mode is off inside forward(); this is what autograd knows)
        plan = _Plan(x, (in_g, in_b), layers, list(reductions), h0, c0,
                     (FLAGS if flags is None else flags) | (0 if needs_bwd else INFERENCE), drop)
        with ops.timed("enc_stack_fwd_T%d_L%d" % (x.shape[1], len(layers))):
            plan.forward()
        hN, cN = plan.final_states()
        out, plan.out = plan.out, None     # the descriptor keeps the raw pointer; no ctx <-> output cycle
        ctx.plan = plan
        ctx.params = params
        ctx.in_norm = (in_g, in_b)
        ctx.mark_non_differentiable(hN, cN)
        return out, hN, cN

    @staticmethod
    def backward(ctx, dout, _dh, _dc):
        plan = ctx.plan
        if plan is None:
            raise RuntimeError("edgedict_amd: the encoder stack's saved activations were consumed "
                               "by a previous backward (retain_graph is not supported)")
        ctx.plan = None
        if dout.dtype != BF16:
            dout = dout.to(BF16)
        with ops.timed("enc_stack_bwd_T%d_L%d" % (plan.T0, plan.L)):
            res = plan.backward(dout, ctx.params, ctx.in_norm)
        if res is None:        # accumulated in place
            return (None,) * (8 + len(ctx.params))
        dig, dib, grads = res
        out = [None, dig, dib, None, None, None, None, None]
        for gb in grads:
            out += [gb["dW_ih"], gb["dW_hh"], gb["db"], gb["db"].clone(), gb["dgamma"], gb["dbeta"]]
        return tuple(out)


def schedule(T0, I0, H, reductions, B=64, chunk=None, backward=False, lag=0, flags=0, grads_final=None):
    """Dry run of the native scheduler (no device): returns ``(step_launch, chunk_enqueued,
    n_launches, max_slots)`` where ``step_launch[l][t]`` is the launch index that carries step t of
    layer l and ``chunk_enqueued[l][k]`` the number of launches issued when chunk k's side-stream
    product was enqueued (-1 = available from the start).  Used by tests/test_stack_schedule.py.
    ``grads_final(layer)`` (backward only) is called where the real pass reports a layer's weight gradients final
    (``edgedict_stack_desc_t.grads_final``): tests/test_dp_gloo.py drives the gradient exchange with it."""
    import numpy as np
    lib = _lib.load()
    L = len(reductions)
    chunk = chunk or CHUNK
    layers = (StackLayer * L)()
    dummy = ctypes.c_void_p(0x1000)          # never dereferenced in a dry run
    fdummy = ctypes.cast(dummy, _fp)
    T, I = T0, I0
    Ts = []
    for l in range(L):
        y = layers[l]
        y.T, y.I, y.reduce, y.residual = T, I, reductions[l], 1 if l > 0 else 0
        for name, typ in StackLayer._fields_[4:]:
            if typ in (_vp, _fp):
                setattr(y, name, fdummy if typ is _fp else dummy)
        y.drop_p, y.drop_seed = 0.0, 0
        Ts.append(T)
        T = (T + reductions[l] - 1) // reductions[l]
        I = H
    d = StackDesc()
    d.B, d.H, d.L, d.chunk, d.lag, d.split_k, d.flags, d.eps = B, H, L, chunk, lag, 0, flags, 1e-5
    d.layers = layers
    d.x, d.x_dtype, d.T0, d.I0 = dummy, 1, T0, I0
    for name in ("in_gamma", "in_beta", "in_mean", "in_rstd", "d_in_gamma", "d_in_beta"):
        setattr(d, name, fdummy)
    d.h0 = d.c0 = None
    d.out = d.dout = d.ws = dummy
    d.ws_bytes = 1 << 62
    cb = None
    if grads_final is not None:
        cb = GRADS_FINAL_CB(lambda layer, _user: grads_final(layer))
        d.grads_final = ctypes.cast(cb, ctypes.c_void_p).value
    f0 = 1
    for r in reductions:
        f0 *= r
    fl, f = [], f0
    for l in range(L):
        fl.append(f)
        f //= reductions[l]
    nch = [(Ts[l] + chunk * fl[l] - 1) // (chunk * fl[l]) for l in range(L)]
    steps = np.full(sum(Ts), -2, dtype=np.int32)
    enq = np.full(sum(nch), -2, dtype=np.int32)
    nl, ms = ctypes.c_int32(0), ctypes.c_int32(0)
    rc = lib.edgedict_stack_schedule(ctypes.byref(d), 1 if backward else 0,
                                     steps.ctypes.data_as(ctypes.c_void_p), enq.ctypes.data_as(ctypes.c_void_p),
                                     ctypes.byref(nl), ctypes.byref(ms))
    _lib.check(rc, "stack_schedule")
    so = np.cumsum([0] + Ts)
    co = np.cumsum([0] + nch)
    return ([steps[so[l]:so[l + 1]] for l in range(L)], [enq[co[l]:co[l + 1]] for l in range(L)],
            int(nl.value), int(ms.value))
