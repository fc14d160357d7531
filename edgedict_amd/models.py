"""MI355X-native RNN-Transducer with the class surface of the reference's ``rnnt/models.py``.

Same constructor arguments, sub-module call signatures, state-dict keys/shapes and return types
as the reference, so ``cli/train.py`` / ``cli/baseline.py`` / ``rnnt/stream.py``
style callers run unchanged — but every piece of arithmetic on the path is a hand-written
gfx950 kernel reached through the C ABI in ``include/edgedict_hip.h``:

=====================  =====================================  ==============================
reference (file:line)  what                                   here
=====================  =====================================  ==============================
rnnt/models.py:131-136 Encoder.forward                        ``Encoder.forward``
rnnt/models.py:55-75   ResLayerNormLSTM.forward               ``_LSTMBlockFn`` per layer
rnnt/models.py:21-29   TimeReduction                          fused in ``layernorm_fwd``
rnnt/models.py:150-157 Decoder.forward                        ``Decoder.forward``
rnnt/models.py:169-179 Joint.forward                          ``_JointFn``
rnnt/models.py:223-241 Transducer.scale_length / forward      ``Transducer``
rnnt/models.py:243-269 Transducer.greedy_decode               ``Transducer.greedy_decode``
=====================  =====================================  ==============================

PyTorch is the container/plumbing layer only: ``nn.Parameter`` holds the fp32 master weights,
``torch.autograd.Function`` routes gradients into ``.grad``, the caching allocator owns memory.
There is no CPU or eager-PyTorch fallback: tensors must live on an MI355X.
"""
import math
import weakref

import torch
from torch import nn

from . import config, encoder_stack, ops, side
from . import _lib
from ._lib import require_cuda
from .loss import _RNNTLossFn
from .tokenizer import BOS, NUL, PAD

F32 = torch.float32


# ----------------------------------------------------------------------------------------
# compute-dtype weight copies (fp32 master -> bf16 / transposed), refreshed when the
# parameter's version counter or the engine's parameter epoch changes
class _WeightCache:
    def __init__(self):
        self._store = {}

    def get(self, p, dtype, transposed=False):
        key = (id(p), dtype, transposed)
        ver = (p.data_ptr(), p._version, config.param_epoch())
        hit = self._store.get(key)
        # ids (and device addresses) are recycled once a tensor dies: the entry must belong to
        # THIS tensor object, not to a dead one that happened to share id, address and version
        if hit is not None and hit[0] == ver and hit[2]() is p:
            return hit[1]
        # a miss: drop the copies of parameters that no longer exist (replicas of nn.DataParallel,
        # rebuilt models) - each holds a device tensor
        dead = [k for k, v in self._store.items() if v[2]() is None]
        for k in dead:
            del self._store[k]
        src = p.detach()
        if not src.is_contiguous():
            src = src.contiguous()
        if transposed == "lstm_fwd":
            t = ops.lstm_pack_weights(src, True, False)[0]
        elif transposed == "lstm_bwd":
            t = ops.lstm_pack_weights(src, False, True)[1]
        elif transposed:
            t = ops.transpose(src, dtype)
        elif dtype == src.dtype:
            t = src
        else:
            t = ops.cast(src, dtype)
        if t is not src and t.is_cuda:
            # the copy is made on whichever stream asked first and read from the other one too (caller's /
            # auxiliary): its block must not be recycled under a reader when the entry is replaced
            from . import side
            for s in (torch.cuda.current_stream(t.device), side.peek(t.device)):
                if s is not None:
                    t.record_stream(s)
        self._store[key] = (ver, t, weakref.ref(p))
        return t

    def clear(self):
        self._store.clear()


WEIGHTS = _WeightCache()


def _lstm_fast(cd, H):
    """bf16 fragment-order recurrence kernels (csrc/lstm_fast.hip) apply?"""
    return cd == torch.bfloat16 and H % 32 == 0 and not config.FORCE_GENERIC_LSTM


def _to_cd(x, cd):
    """Bring an activation into the compute dtype (device cast kernel, no-op if already there)."""
    if x.dtype == cd:
        return x.contiguous()
    if x.dtype not in (torch.float32, torch.bfloat16):
        x = x.float()
    return ops.cast(x.contiguous(), cd)


def _state(t):
    """Recurrent state slice -> contiguous fp32 [B,H] (or None)."""
    if t is None:
        return None
    t = t.detach()
    if t.dtype != F32:
        t = ops.cast(t.contiguous(), F32)
    return t.contiguous()


def _grads_ready(params, device):
    """Tell the data-parallel gradient exchange (dp.BucketedAllReduce.ready) that ``params`` were
    accumulated in place by work enqueued on the auxiliary stream up to now."""
    from . import dp
    if dp.READY_HOOK is not None:
        dp.READY_HOOK(params, side.stream(device))


# ----------------------------------------------------------------------------------------
# autograd plumbing around the kernels
class _InputNormFn(torch.autograd.Function):
    """LayerNorm(input_size) on the stacked log-mel input (rnnt/models.py:124,132)."""

    @staticmethod
    def forward(ctx, xs, gamma, beta, cd):
        x = _to_cd(xs, cd)
        y, mean, rstd = ops.layernorm_fwd(x, None, gamma.detach(), beta.detach(), 1)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.in_dtype = xs.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        ds, dgamma, dbeta = ops.layernorm_bwd(dy, x, None, gamma.detach(), mean, rstd, 1)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ds if ds.dtype == ctx.in_dtype else ops.cast(ds, ctx.in_dtype)
        return dx, dgamma, dbeta, None


class _LSTMBlockFn(torch.autograd.Function):
    """One 1-layer LSTM over the whole sequence, optionally followed by the encoder's
    residual add + LayerNorm (+ TimeReduction).

    forward : G = x W_ih^T + b_ih + b_hh (one MFMA GEMM) -> T step kernels -> fused LN epilogue
    backward: LN backward -> T BPTT step kernels (G becomes dG in place) -> dX, dW_ih, dW_hh as
              GEMMs reading dG / x / h_{t-1} transposed in place, db as a column sum.
    """

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, ln_w, ln_b, h0, c0, residual, reduce, cd):
        B, T, I = x.shape
        H = w_hh.shape[1]
        wih = WEIGHTS.get(w_ih, cd)
        fast = _lstm_fast(cd, H)
        whh = None if fast else WEIGHTS.get(w_hh, cd)
        whh_p = WEIGHTS.get(w_hh, cd, "lstm_fwd") if fast else None
        G = ops.gemm(x.view(B * T, I), wih, bias=b_ih.detach(), bias2=b_hh.detach())
        G = G.view(B, T, 4 * H)
        with ops.timed("lstm_fwd_T%d_H%d" % (T, H)):
            Y, Hprev, Cst, hN, cN = ops.lstm_forward(G, whh, h0, c0, whh_p)
        if ln_w is not None:
            out, mean, rstd = ops.layernorm_fwd(Y, x if residual else None, ln_w.detach(),
                                                ln_b.detach(), reduce)
        else:
            out, mean, rstd = Y, None, None
        ctx.save_for_backward(x, w_ih, w_hh, ln_w, c0)
        ctx.biases = (b_ih, b_hh)
        ctx.inter = (G, Y, Hprev, Cst, mean, rstd)
        ctx.cfg = (residual, reduce, cd, ln_w is not None)
        ctx.mark_non_differentiable(hN, cN)
        return out, hN, cN

    @staticmethod
    def backward(ctx, dout, _dh, _dc):
        if ctx.inter is None:
            raise RuntimeError("edgedict_amd: this LSTM block's saved gates were consumed by a "
                               "previous backward (retain_graph is not supported)")
        x, w_ih, w_hh, ln_w, c0 = ctx.saved_tensors
        G, Y, Hprev, Cst, mean, rstd = ctx.inter
        ctx.inter = None
        residual, reduce, cd, has_ln = ctx.cfg
        B, T, I = x.shape
        H = w_hh.shape[1]
        dgamma = dbeta = None
        if has_ln:
            ds, dgamma, dbeta = ops.layernorm_bwd(dout, Y, x if residual else None,
                                                  ln_w.detach(), mean, rstd, reduce)
        else:
            ds = dout.contiguous()
        with ops.timed("lstm_bwd_T%d_H%d" % (T, H)):
            if _lstm_fast(cd, H):
                ops.lstm_backward(G, ds, Cst, c0, None, WEIGHTS.get(w_hh, cd, "lstm_bwd"))
            else:
                ops.lstm_backward(G, ds, Cst, c0, WEIGHTS.get(w_hh, cd, transposed=True))
        dG = G.view(B * T, 4 * H)
        x2 = x.view(B * T, I)
        M = B * T
        # the layer below waits for dx only: it goes first; the weight gradients feed nothing downstream
        dx = None
        if ctx.needs_input_grad[0]:
            wih = WEIGHTS.get(w_ih, cd)
            if residual and has_ln:
                dx = ds  # d(residual) + dG W_ih, accumulated in place by the GEMM epilogue
                ops.gemm(dG, wih.t(), out=ds.view(M, I), accumulate=True)
            else:
                dx = ops.gemm(dG, wih.t()).view(B, T, I)
        b_ih, b_hh = ctx.biases
        if config.DEFER_WEIGHT_GRADS and config.DEFER_LSTM_WEIGHT_GRADS and all(
                p.grad is not None and p.grad.dtype == F32 and p.grad.is_contiguous()
                for p in (w_ih, w_hh, b_ih, b_hh)):
            # ... so they accumulate straight into the .grad buffers on the auxiliary stream, under the next
            # layer's BPTT (as the joint's dW2 does, side.py)
            with side.deferred(dG.device, G, x, Hprev):
                ops.gemm(dG.t(), x2.t(), out=w_ih.grad, accumulate=True, split_k=ops.pick_split_k(4 * H, I, M))
                ops.gemm(dG.t(), Hprev.view(M, H).t(), out=w_hh.grad, accumulate=True,
                         split_k=ops.pick_split_k(4 * H, H, M))
                ops.colsum(dG, out=b_ih.grad)
                ops.colsum(dG, out=b_hh.grad)
            _grads_ready((w_ih, w_hh, b_ih, b_hh), dG.device)      # no autograd hook fires for them (dp.py)
            return (dx, None, None, None, None, dgamma, dbeta, None, None, None, None, None)
        dw_ih = ops.gemm(dG.t(), x2.t(), out_dtype=F32, split_k=ops.pick_split_k(4 * H, I, M))
        dw_hh = ops.gemm(dG.t(), Hprev.view(M, H).t(), out_dtype=F32,
                         split_k=ops.pick_split_k(4 * H, H, M))
        db = ops.colsum(dG)
        return (dx, dw_ih, dw_hh, db, db.clone(), dgamma, dbeta, None, None, None, None, None)


class _GRUBlockFn(torch.autograd.Function):
    """One 1-layer GRU over the whole sequence + the encoder's residual add / LayerNorm /
    TimeReduction, as ``_LSTMBlockFn`` (ResLayerNormGRU.forward, rnnt/models.py:99-116).

    forward : G = x W_ih^T + b_ih (one MFMA GEMM) -> T step kernels (csrc/gru.hip) -> fused LN epilogue
    backward: LN backward -> T BPTT step kernels (G -> input-side, DH -> hidden-side pre-activation
              gradients) -> dX, dW_ih, dW_hh as GEMMs, both biases as column sums."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, ln_w, ln_b, h0, residual, reduce, cd):
        B, T, I = x.shape
        H = w_hh.shape[1]
        G = ops.gemm(x.view(B * T, I), WEIGHTS.get(w_ih, cd), bias=b_ih.detach()).view(B, T, 3 * H)
        with ops.timed("gru_fwd_T%d_H%d" % (T, H)):
            Y, Hprev, HN, hN = ops.gru_forward(G, WEIGHTS.get(w_hh, cd), b_hh.detach(), h0)
        if ln_w is not None:
            out, mean, rstd = ops.layernorm_fwd(Y, x if residual else None, ln_w.detach(),
                                                ln_b.detach(), reduce)
        else:
            out, mean, rstd = Y, None, None
        ctx.save_for_backward(x, w_ih, w_hh, ln_w)
        ctx.inter = (G, Y, Hprev, HN, mean, rstd)
        ctx.cfg = (residual, reduce, cd, ln_w is not None)
        ctx.mark_non_differentiable(hN)
        return out, hN

    @staticmethod
    def backward(ctx, dout, _dh):
        if ctx.inter is None:
            raise RuntimeError("edgedict_amd: this GRU block's saved gates were consumed by a "
                               "previous backward (retain_graph is not supported)")
        x, w_ih, w_hh, ln_w = ctx.saved_tensors
        G, Y, Hprev, HN, mean, rstd = ctx.inter
        ctx.inter = None
        residual, reduce, cd, has_ln = ctx.cfg
        B, T, I = x.shape
        H = w_hh.shape[1]
        dgamma = dbeta = None
        if has_ln:
            ds, dgamma, dbeta = ops.layernorm_bwd(dout, Y, x if residual else None,
                                                  ln_w.detach(), mean, rstd, reduce)
        else:
            ds = dout.contiguous()
        with ops.timed("gru_bwd_T%d_H%d" % (T, H)):
            DH = ops.gru_backward(G, ds, Hprev, HN, WEIGHTS.get(w_hh, cd, transposed=True))
        M = B * T
        dG, dH2, x2 = G.view(M, 3 * H), DH.view(M, 3 * H), x.view(M, I)
        dw_ih = ops.gemm(dG.t(), x2.t(), out_dtype=F32, split_k=ops.pick_split_k(3 * H, I, M))
        dw_hh = ops.gemm(dH2.t(), Hprev.view(M, H).t(), out_dtype=F32,
                         split_k=ops.pick_split_k(3 * H, H, M))
        db_ih, db_hh = ops.colsum(dG), ops.colsum(dH2)
        dx = None
        if ctx.needs_input_grad[0]:
            wih = WEIGHTS.get(w_ih, cd)
            if residual and has_ln:
                dx = ds  # d(residual) + dG W_ih, accumulated in place by the GEMM epilogue
                ops.gemm(dG, wih.t(), out=ds.view(M, I), accumulate=True)
            else:
                dx = ops.gemm(dG, wih.t()).view(B, T, I)
        return (dx, dw_ih, dw_hh, db_ih, db_hh, dgamma, dbeta, None, None, None, None)


class _DropoutFn(torch.autograd.Function):
    """Training-mode dropout with a counter-based mask (csrc/elementwise.hip): the backward pass
    re-applies the same (p, seed) to the gradient, no mask tensor is kept."""

    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.cfg = (float(p), int(seed))
        return ops.dropout(x, p, seed)

    @staticmethod
    def backward(ctx, dy):
        p, seed = ctx.cfg
        return ops.dropout(dy.contiguous(), p, seed), None, None


_dropout_calls = [0]


def _next_dropout_seed():
    """A fresh mask per call: seed = torch's RNG stream (so torch.manual_seed governs it) mixed with
    a call counter."""
    _dropout_calls[0] += 1
    return (torch.initial_seed() * 0x9E3779B1 + _dropout_calls[0] * 0x85EBCA6B) & 0xFFFFFFFF


def _dropout(x, p):
    return _DropoutFn.apply(x, p, _next_dropout_seed())


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b on the last dimension (nn.Linear: rnnt/models.py:129,135,148,156)."""

    @staticmethod
    def forward(ctx, x, w, b, cd):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        wc = WEIGHTS.get(w, cd)
        y = ops.gemm(x2, wc, bias=b.detach() if b is not None else None)
        ctx.save_for_backward(x2, w)
        ctx.bias = b
        ctx.cfg = (cd, shp, b is not None)
        return y.view(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        cd, shp, has_b = ctx.cfg
        b = ctx.bias
        dy2 = dy.reshape(-1, w.shape[0])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        M = x2.shape[0]
        # dx first: it is what the rest of the backward pass waits for (the encoder's projection sits between the joint
        # and the stack's BPTT); dW / db go to the auxiliary stream when they can accumulate in place (side.py)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm(dy2, WEIGHTS.get(w, cd).t()).view(*shp)
        if config.DEFER_WEIGHT_GRADS and config.DEFER_LSTM_WEIGHT_GRADS and all(
                p is None or (p.grad is not None and p.grad.dtype == F32 and p.grad.is_contiguous()) for p in (w, b)):
            with side.deferred(dy2.device, dy2, x2):
                ops.gemm(dy2.t(), x2.t(), out=w.grad, accumulate=True,
                         split_k=ops.pick_split_k(w.shape[0], w.shape[1], M))
                if has_b:
                    ops.colsum(dy2, out=b.grad)
            _grads_ready((w, b) if has_b else (w,), dy2.device)     # no autograd hook fires for them (dp.py)
            return dx, None, None, None
        dw = ops.gemm(dy2.t(), x2.t(), out_dtype=F32,
                      split_k=ops.pick_split_k(w.shape[0], w.shape[1], M))
        db = ops.colsum(dy2) if has_b else None
        return dx, dw, db, None


class _EmbeddingFn(torch.autograd.Function):
    """BOS left-pad + nn.Embedding(padding_idx=PAD) (rnnt/models.py:150-153)."""

    @staticmethod
    def forward(ctx, tokens, weight, prepend_bos, cd):
        out = ops.embedding_fwd(tokens, weight.detach(), cd, prepend_bos, BOS)
        ctx.save_for_backward(tokens)
        ctx.cfg = (weight.shape[0], prepend_bos)
        return out

    @staticmethod
    def backward(ctx, dout):
        (tokens,) = ctx.saved_tensors
        V, prepend_bos = ctx.cfg
        return None, ops.embedding_bwd(tokens, dout, V, prepend_bos, BOS, PAD), None, None


class _JointFn(torch.autograd.Function):
    """logits[b,t,u,:] = W2 tanh(W1 [enc[b,t]; dec[b,u]] + b1) + b2 (rnnt/models.py:169-179).

    W1 is applied as two small GEMMs (W1[:, :P_enc] on enc, W1[:, P_enc:] on dec) followed by a
    broadcast-add+tanh kernel, so the reference's [B,T,U+1,P_enc+P_dec] concat is never built.
    """

    @staticmethod
    def forward(ctx, enc, dec, w1, b1, w2, b2, cd):
        B, T, P = enc.shape
        U1, P2 = dec.shape[1], dec.shape[2]
        J, V = w1.shape[0], w2.shape[0]
        w1c = WEIGHTS.get(w1, cd)
        w2c = WEIGHTS.get(w2, cd)
        enc2 = enc.reshape(B * T, P)
        dec2 = dec.reshape(B * U1, P2)
        E1 = ops.gemm(enc2, w1c[:, :P])
        D1 = ops.gemm(dec2, w1c[:, P:], bias=b1.detach())
        hid = ops.joint_hidden_fwd(E1.view(B, T, J), D1.view(B, U1, J))
        with ops.timed("joint_logits_gemm"):
            logits = ops.gemm(hid.view(B * T * U1, J), w2c, bias=b2.detach())
        ctx.save_for_backward(enc2, dec2, w1, w2, hid)
        ctx.b1, ctx.b2 = b1, b2
        ctx.cfg = (cd, B, T, U1, P, P2, J, V)
        return logits.view(B, T, U1, V)

    @staticmethod
    def backward(ctx, dlogits):
        ops.mark("joint_bwd:enter")
        enc2, dec2, w1, w2, hid = ctx.saved_tensors
        cd, B, T, U1, P, P2, J, V = ctx.cfg
        M = B * T * U1
        dl = dlogits.reshape(M, V)
        if not dl.is_contiguous():
            dl = dl.contiguous()
        hid2 = hid.view(M, J)
        w1c = WEIGHTS.get(w1, cd)
        w2c = WEIGHTS.get(w2, cd)
        # weight gradients feed nothing downstream: when the parameters already own fp32 .grad
        # buffers they are accumulated in place on the auxiliary stream, under the encoder's
        # backward pass (side.py); otherwise they are returned to autograd as usual
        defer = config.DEFER_WEIGHT_GRADS and all(
            p.grad is not None and p.grad.dtype == F32 and p.grad.is_contiguous()
            for p in (w1, ctx.b1, w2, ctx.b2))
        dw1 = db1 = dw2 = db2 = None
        with ops.timed("joint_dhid_gemm"):
            # dl x W2 with W2^T materialised once per optimiser step (1.3 M elements): both
            # operands K-contiguous -> the direct-to-LDS kernel (gemm_nt.hip)
            dhid = ops.gemm(dl, WEIGHTS.get(w2, cd, transposed=True))
        if not defer:
            with ops.timed("joint_dw2_gemm"):
                dw2 = ops.gemm(dl.t(), hid2.t(), out_dtype=F32, split_k=ops.pick_split_k(V, J, M))
            db2 = ops.colsum(dl)
        dE1, dD1 = ops.joint_hidden_bwd(dhid.view(B, T, U1, J), hid)
        del dhid
        dE1c = ops.cast(dE1, cd).view(B * T, J)
        dD1c = ops.cast(dD1, cd).view(B * U1, J)
        denc = ops.gemm(dE1c, w1c[:, :P].t()).view(B, T, P)
        ddec = ops.gemm(dD1c, w1c[:, P:].t()).view(B, U1, P2)
        if defer:
            # enqueued AFTER the critical-path products above: the auxiliary stream starts when
            # they are done and its MFMA work runs under the latency-bound recurrences that follow
            with side.deferred(dl.device, dl, hid, dE1c, dD1c, dD1, enc2, dec2):
                ops.gemm(dl.t(), hid2.t(), out=w2.grad, accumulate=True, split_k=8,
                         max_wg_per_cu=2)
                ops.colsum(dl, out=ctx.b2.grad)
                g1 = w1.grad
                ops.gemm(dE1c.t(), enc2.t(), out=g1[:, :P], accumulate=True,
                         split_k=ops.pick_split_k(J, P, B * T))
                ops.gemm(dD1c.t(), dec2.t(), out=g1[:, P:], accumulate=True,
                         split_k=ops.pick_split_k(J, P2, B * U1))
                ops.colsum(dD1.view(B * U1, J), out=ctx.b1.grad)
            _grads_ready((w1, ctx.b1, w2, ctx.b2), dl.device)
        if not defer:
            dw1 = torch.empty(J, P + P2, dtype=F32, device=dl.device)
            ops.gemm(dE1c.t(), enc2.t(), out=dw1[:, :P], split_k=ops.pick_split_k(J, P, B * T))
            ops.gemm(dD1c.t(), dec2.t(), out=dw1[:, P:], split_k=ops.pick_split_k(J, P2, B * U1))
            db1 = ops.colsum(dD1.view(B * U1, J))
        ops.mark("joint_bwd:exit")
        return denc, ddec, dw1, db1, dw2, db2, None


class _JointLossFn(torch.autograd.Function):
    """Joint network + RNN-T loss on the PACKED lattice (training path of Transducer.forward).

    Same arithmetic as ``_JointFn`` followed by ``_RNNTLossFn`` (rnnt/models.py:169-179,
    221,238), but only the cells inside each utterance's (T_b, U_b+1) box exist: hid, logits and
    their gradients are [M_valid, .] matrices (row of (b,t,u) = off[b] + t (U_b+1) + u).  Cells
    outside the box have zero gradient and no influence on the loss, so nothing changes
    numerically; the two big products and the loss kernels just do not touch padding
    (35 % of the rows on the bench batch).  Needs the lengths on the HOST (they size M_valid)."""

    @staticmethod
    def forward(ctx, enc, dec, w1, b1, w2, b2, labels, act_lens, label_lens, blank, cd):
        from ._staging import to_device
        B, T, P = enc.shape
        U1, P2 = dec.shape[1], dec.shape[2]
        J, V = w1.shape[0], w2.shape[0]
        dev = enc.device
        al = act_lens.to(torch.int64).cpu()
        ll = label_lens.to(torch.int64).cpu()
        if int(al.max()) != T or int(ll.max()) != U1 - 1 or int(al.min()) < 1 or int(ll.min()) < 0:
            raise ValueError("Input length mismatch")     # wording of warprnnt_pytorch's checks
        rows = al * (ll + 1)
        off = torch.zeros(B, dtype=torch.int64)
        off[1:] = torch.cumsum(rows, 0)[:-1]
        M = int(rows.sum())
        off_d = to_device(off, dev)
        al_d = to_device(al.to(torch.int32), dev)
        ll_d = to_device(ll.to(torch.int32), dev)
        w1c = WEIGHTS.get(w1, cd)
        w2c = WEIGHTS.get(w2, cd)
        enc2 = enc.reshape(B * T, P)
        dec2 = dec.reshape(B * U1, P2)
        E1 = ops.gemm(enc2, w1c[:, :P])
        D1 = ops.gemm(dec2, w1c[:, P:], bias=b1.detach())
        hid = torch.empty(M, J, dtype=cd, device=dev)
        with ops.timed("joint_hidden_fwd"):
            _lib.call("joint_hidden_fwd_packed", _lib.dtype_code(cd), E1, D1, hid, al_d, ll_d, off_d,
                      B, T, U1, J)
        ops.LAST["joint_rows"] = M
        lib = _lib.load()
        ws = torch.empty(lib.edgedict_rnnt_workspace_bytes(B, T, U1), dtype=torch.uint8, device=dev)
        costs = torch.empty(B, dtype=F32, device=dev)
        reduced = torch.empty(1, dtype=F32, device=dev)
        if config.FUSED_LSE and cd == torch.bfloat16 and J >= 128 and J % 64 == 0 and V % 8 == 0 and M >= 256:
            # logits product with the log-softmax partials in its epilogue (gemm_nt256.hip): the loss
            # finishes the denominators from M x V/64 pairs instead of re-reading the logits
            slots = (V + 63) // 64
            parts = torch.empty(M, slots, 2, dtype=F32, device=dev)
            logits = torch.empty(M, V, dtype=cd, device=dev)
            # (TrainEngine hangs the next batch's front-end here: beside this matrix-bound product it is nearly free;
            # beside the encoder's recurrence, or beside the bandwidth-bound tanh kernel above, it cost what it saved)
            ops.fire("before_logits_gemm")
            with ops.timed("joint_logits_gemm"):
                _lib.call("gemm_nt_lse", hid, ops._ll(J), w2c, ops._ll(J), logits, ops._ll(V), M, V, J,
                          b2.detach(), parts)
            with ops.timed("rnnt_loss_fwd"):
                _lib.call("rnnt_loss_forward_packed_parts", logits, labels, al_d, ll_d, off_d, B, T, U1, V,
                          int(blank), costs, reduced, 1.0 / B, ws, parts, slots)
        else:
            with ops.timed("joint_logits_gemm"):
                logits = ops.gemm(hid, w2c, bias=b2.detach())
            _lib.call("rnnt_loss_forward_packed", logits, _lib.dtype_code(cd), labels, al_d, ll_d, off_d,
                      B, T, U1, V, int(blank), costs, reduced, 1.0 / B, ws)
        ops.LAST["joint_costs"] = costs       # per-utterance costs of the last packed joint + loss ([B], device)
        ctx.save_for_backward(enc2, dec2, w1, w2, hid, logits, labels, al_d, ll_d, off_d, ws)
        ctx.b1, ctx.b2 = b1, b2
        ctx.cfg = (cd, B, T, U1, P, P2, J, V, M, int(blank))
        return reduced

    @staticmethod
    def backward(ctx, gout):
        ops.mark("joint_bwd:enter")
        enc2, dec2, w1, w2, hid, logits, labels, al_d, ll_d, off_d, ws = ctx.saved_tensors
        cd, B, T, U1, P, P2, J, V, M, blank = ctx.cfg
        dl = torch.empty_like(logits)
        gscale = gout.contiguous().float()
        w1c = WEIGHTS.get(w1, cd)
        w2t = WEIGHTS.get(w2, cd, transposed=True)
        defer = config.DEFER_WEIGHT_GRADS and all(
            p.grad is not None and p.grad.dtype == F32 and p.grad.is_contiguous()
            for p in (w1, ctx.b1, w2, ctx.b2))
        dw1 = db1 = dw2 = db2 = None
        dE1 = torch.empty(B, T, J, dtype=F32, device=dl.device)
        dD1 = torch.empty(B, U1, J, dtype=F32, device=dl.device)
        # (pipelining the loss gradient against the dhid product by utterance groups on two streams was measured:
        # 21.70 ms per step in one pass, 21.74 / 22.26 / 23.18 with 2 / 4 / 8 groups - both sit on the L2/HBM path)
        # the output bias's gradient is the column sum of dl: the loss-gradient kernel leaves it as per-workgroup
        # partial rows (a few MB) instead of a second pass over dl beside the encoder's BPTT (0.4 ms per step)
        cs_rows = _lib.load().edgedict_rnnt_grad_colsum_rows(_lib.dtype_code(cd), B, T, U1, V) if config.FUSED_DB2 else 0
        db2_parts = torch.empty(cs_rows, V, dtype=F32, device=dl.device) if cs_rows > 0 else None
        with ops.timed("rnnt_grad"):
            if db2_parts is not None:
                _lib.call("rnnt_loss_backward_packed_colsum", logits, _lib.dtype_code(cd), dl, labels, al_d, ll_d,
                          off_d, B, T, U1, V, blank, ws, 1.0 / B, gscale, 0, db2_parts)
            else:
                _lib.call("rnnt_loss_backward_packed", logits, _lib.dtype_code(cd), dl, labels, al_d, ll_d,
                          off_d, B, T, U1, V, blank, ws, 1.0 / B, gscale, 0)
        del logits
        with ops.timed("joint_dhid_gemm"):
            dhid = ops.gemm(dl, w2t)
        with ops.timed("joint_hidden_bwd"):
            _lib.call("joint_hidden_bwd_packed", _lib.dtype_code(cd), dhid, hid, dE1, dD1, al_d, ll_d,
                      off_d, B, T, U1, J)
        if not defer:
            dw2 = ops.gemm(dl.t(), hid.t(), out_dtype=F32, split_k=ops.pick_split_k(V, J, M))
            db2 = ops.colsum(dl if db2_parts is None else db2_parts)
        del dhid
        dE1c = ops.cast(dE1, cd).view(B * T, J)
        dD1c = ops.cast(dD1, cd).view(B * U1, J)
        denc = ops.gemm(dE1c, w1c[:, :P].t()).view(B, T, P)
        ddec = ops.gemm(dD1c, w1c[:, P:].t()).view(B, U1, P2)
        if defer:
            with side.deferred(dl.device, dl, hid, dE1c, dD1c, dD1, enc2, dec2, db2_parts):
                ops.gemm(dl.t(), hid.t(), out=w2.grad, accumulate=True, split_k=8, max_wg_per_cu=2)
                ops.colsum(dl if db2_parts is None else db2_parts, out=ctx.b2.grad)
                g1 = w1.grad
                ops.gemm(dE1c.t(), enc2.t(), out=g1[:, :P], accumulate=True,
                         split_k=ops.pick_split_k(J, P, B * T))
                ops.gemm(dD1c.t(), dec2.t(), out=g1[:, P:], accumulate=True,
                         split_k=ops.pick_split_k(J, P2, B * U1))
                ops.colsum(dD1.view(B * U1, J), out=ctx.b1.grad)
            _grads_ready((w1, ctx.b1, w2, ctx.b2), dl.device)
        else:
            dw1 = torch.empty(J, P + P2, dtype=F32, device=dl.device)
            ops.gemm(dE1c.t(), enc2.t(), out=dw1[:, :P], split_k=ops.pick_split_k(J, P, B * T))
            ops.gemm(dD1c.t(), dec2.t(), out=dw1[:, P:], split_k=ops.pick_split_k(J, P2, B * U1))
            db1 = ops.colsum(dD1.view(B * U1, J))
        ops.mark("joint_bwd:exit")
        return denc, ddec, dw1, db1, dw2, db2, None, None, None, None, None


# ----------------------------------------------------------------------------------------
# parameter containers with the reference's names, shapes and default initialisation
class TimeReduction(nn.Module):
    """Marker for the 2x time reduction of rnnt/models.py:16-29; the arithmetic (zero-pad to an
    even length after the LayerNorm, mean of frame pairs) is fused into the LayerNorm kernel."""

    def __init__(self, reduction_factor=2):
        super().__init__()
        self.reduction_factor = reduction_factor


class _LayerNormParams(nn.Module):
    def __init__(self, size, eps=1e-5):
        super().__init__()
        self.normalized_shape = (size,)
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(size))
        self.bias = nn.Parameter(torch.zeros(size))


class _LinearParams(nn.Module):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        # nn.Linear default init
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(in_features)
        nn.init.uniform_(self.bias, -bound, bound)


class _LSTMParams(nn.Module):
    """nn.LSTM-compatible parameter set: weight_ih_l{k} [4H,I], weight_hh_l{k} [4H,H],
    bias_ih_l{k}, bias_hh_l{k} [4H]; gate order i,f,g,o; uniform(-1/sqrt(H), 1/sqrt(H))."""

    def __init__(self, input_size, hidden_size, num_layers=1, dropout=0.0):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.num_layers, self.dropout = num_layers, dropout
        k = 1.0 / math.sqrt(hidden_size)
        for layer in range(num_layers):
            i = input_size if layer == 0 else hidden_size
            for name, shape in (("weight_ih", (4 * hidden_size, i)),
                                ("weight_hh", (4 * hidden_size, hidden_size)),
                                ("bias_ih", (4 * hidden_size,)),
                                ("bias_hh", (4 * hidden_size,))):
                p = nn.Parameter(torch.empty(*shape).uniform_(-k, k))
                setattr(self, "%s_l%d" % (name, layer), p)

    def layer(self, k):
        return tuple(getattr(self, "%s_l%d" % (n, k))
                     for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))

    def flatten_parameters(self):  # API compatibility (cuDNN concept; nothing to do here)
        pass


class ResLayerNormLSTM(nn.Module):
    """Stack of 1-layer LSTMs with residual connections, LayerNorm and optional time reduction
    (reference rnnt/models.py:32-75).  Padded frames are NOT masked, as in the reference."""

    def __init__(self, input_size, hidden_size, num_layers, dropout=0,
                 time_reductions=[1], reduction_factor=2):
        super().__init__()
        if reduction_factor != 2:
            raise ValueError("only reduction_factor=2 is implemented (the reference default)")
        self.dropout = dropout      # nn.Dropout after LayerNorm(+TimeReduction), rnnt/models.py:47-53
        self.hidden_size = hidden_size
        self.lstms = nn.ModuleList()
        self.projs = nn.ModuleList()
        self.reductions = []
        for i in range(num_layers):
            self.lstms.append(_LSTMParams(input_size, hidden_size, 1))
            proj = [_LayerNormParams(hidden_size)]
            if i in time_reductions:
                proj.append(TimeReduction(reduction_factor))
            self.reductions.append(2 if i in time_reductions else 1)
            input_size = hidden_size
            self.projs.append(nn.Sequential(*proj))

    def forward(self, xs, hiddens=None, cd=None):
        cd = cd or config.get_compute_dtype()
        xs = _to_cd(xs, cd) if xs.dtype != cd else xs
        new_hs, new_cs = [], []
        for i, (lstm, proj) in enumerate(zip(self.lstms, self.projs)):
            h0 = c0 = None
            if hiddens is not None:
                h0, c0 = _state(hiddens[0][i]), _state(hiddens[1][i])
            w_ih, w_hh, b_ih, b_hh = lstm.layer(0)
            xs, h, c = _LSTMBlockFn.apply(xs.contiguous(), w_ih, w_hh, b_ih, b_hh,
                                          proj[0].weight, proj[0].bias, h0, c0, i != 0,
                                          self.reductions[i], cd)
            if self.dropout > 0 and self.training:
                xs = _dropout(xs, self.dropout)
            new_hs.append(h)
            new_cs.append(c)
        return xs, (torch.stack(new_hs, 0), torch.stack(new_cs, 0))


class _GRUParams(nn.Module):
    """nn.GRU-compatible 1-layer parameter set: weight_ih_l0 [3H,I], weight_hh_l0 [3H,H],
    bias_ih_l0, bias_hh_l0 [3H]; gate order r,z,n; uniform(-1/sqrt(H), 1/sqrt(H))."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, 1
        k = 1.0 / math.sqrt(hidden_size)
        for name, shape in (("weight_ih", (3 * hidden_size, input_size)),
                            ("weight_hh", (3 * hidden_size, hidden_size)),
                            ("bias_ih", (3 * hidden_size,)), ("bias_hh", (3 * hidden_size,))):
            setattr(self, name + "_l0", nn.Parameter(torch.empty(*shape).uniform_(-k, k)))

    def layer(self, k=0):
        return tuple(getattr(self, n + "_l0") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))

    def flatten_parameters(self):
        pass


class ResLayerNormGRU(nn.Module):
    """Stack of 1-layer GRUs with residual connections, LayerNorm and optional time reduction
    (reference rnnt/models.py:77-116; the module list is called ``lstms`` there too, so the
    state-dict keys are those of the LSTM variant with 3H rows).  ``hiddens`` is ONE tensor
    [L,B,H].  Per-layer kernels only (csrc/gru.hip): this variant is a compatibility module."""

    def __init__(self, input_size, hidden_size, num_layers, dropout=0,
                 time_reductions=[1], reduction_factor=2):
        super().__init__()
        if reduction_factor != 2:
            raise ValueError("only reduction_factor=2 is implemented (the reference default)")
        self.dropout = dropout
        self.hidden_size = hidden_size
        self.lstms = nn.ModuleList()
        self.projs = nn.ModuleList()
        self.reductions = []
        for i in range(num_layers):
            self.lstms.append(_GRUParams(input_size, hidden_size))
            proj = [_LayerNormParams(hidden_size)]
            if i in time_reductions:
                proj.append(TimeReduction(reduction_factor))
            self.reductions.append(2 if i in time_reductions else 1)
            input_size = hidden_size
            self.projs.append(nn.Sequential(*proj))

    def forward(self, xs, hiddens=None, cd=None):
        cd = cd or config.get_compute_dtype()
        xs = _to_cd(xs, cd) if xs.dtype != cd else xs
        new_hs = []
        for i, (gru, proj) in enumerate(zip(self.lstms, self.projs)):
            h0 = _state(hiddens[i]) if hiddens is not None else None
            w_ih, w_hh, b_ih, b_hh = gru.layer(0)
            xs, h = _GRUBlockFn.apply(xs.contiguous(), w_ih, w_hh, b_ih, b_hh, proj[0].weight,
                                      proj[0].bias, h0, i != 0, self.reductions[i], cd)
            if self.dropout > 0 and self.training:
                xs = _dropout(xs, self.dropout)
            new_hs.append(h)
        return xs, torch.stack(new_hs, 0)


class Encoder(nn.Module):
    """LayerNorm -> ResLayerNormLSTM -> Linear (reference rnnt/models.py:119-136)."""

    def __init__(self, input_size, hidden_size, num_layers, dropout, proj_size,
                 module=ResLayerNormLSTM, time_reductions=[1], has_proj=True):
        super().__init__()
        self.norm = _LayerNormParams(input_size)
        self.lstm = module(input_size, hidden_size, num_layers, dropout=dropout,
                           time_reductions=time_reductions)
        self.has_proj = has_proj
        if has_proj:
            self.proj = _LinearParams(hidden_size, proj_size)

    def forward(self, xs, hiddens=None):
        require_cuda(xs)
        cd = getattr(self, "compute_dtype", None) or config.get_compute_dtype()
        lstm = self.lstm
        drop = getattr(lstm, "dropout", 0) > 0 and self.training    # applied inside the stack / by the per-layer path
        # short inputs (streaming chunks of a few frames) stay on the per-layer kernels: the
        # wavefront needs ~5 lags of launches to fill, more than 6 x T per-layer steps for small T
        if (isinstance(lstm, ResLayerNormLSTM) and xs.dim() == 3
                and xs.shape[1] >= config.STACK_MIN_FRAMES
                and encoder_stack.supported(cd, lstm.hidden_size, xs.shape[2], len(lstm.lstms),
                                            lstm.reductions)):
            # bf16: input LayerNorm + all layers as one layer-pipelined native call per direction
            h0 = c0 = None
            if hiddens is not None:
                h0 = _state(hiddens[0]).contiguous()
                c0 = _state(hiddens[1]).contiguous()
            params = []
            for m, proj in zip(lstm.lstms, lstm.projs):
                params += list(m.layer(0)) + [proj[0].weight, proj[0].bias]
            # nn.Dropout behind every layer's LayerNorm (+ TimeReduction), rnnt/models.py:47-53,70: inside the stack's
            # norm role, one seed per layer drawn in the order the per-layer path draws them
            dcfg = (float(lstm.dropout), tuple(_next_dropout_seed() for _ in lstm.lstms)) if drop else None
            xs, h, c = encoder_stack.EncoderStackFn.apply(
                xs, self.norm.weight, self.norm.bias, h0, c0, tuple(lstm.reductions), None, dcfg, *params)
            hiddens = (h, c)
        elif (isinstance(lstm, ResLayerNormLSTM) and xs.dim() == 3 and cd == torch.bfloat16 and not drop
              and not torch.is_grad_enabled() and 0 < xs.shape[1] < config.STACK_MIN_FRAMES
              and xs.shape[0] <= (config.STREAM_STEP_MAX_ROWS_SHORT if xs.shape[1] <= 2 else config.STREAM_STEP_MAX_ROWS)
              and lstm.hidden_size % 32 == 0 and xs.shape[2] % 8 == 0 and config.STREAM_ENCODER_STEP):
            # streaming chunks of a FEW streams (rnnt/stream.py:93-100: a frame or two per call): ONE native call, a
            # fused launch per layer-frame (csrc/decode_fused.hip, edgedict_stream_encoder_step) - 0.39 instead of
            # 0.67 ms per chunk for one stream, 0.33 instead of 0.61 for 256; longer chunks of many streams stay on the
            # per-layer kernels, which batch the input product over the frames (config.STREAM_STEP_MAX_ROWS*)
            xs, hiddens = _stream_encoder_step(self, xs, hiddens)
        else:
            xs = _InputNormFn.apply(xs, self.norm.weight, self.norm.bias, cd)
            xs, hiddens = self.lstm(xs, hiddens, cd)
        if self.has_proj:
            xs = _LinearFn.apply(xs, self.proj.weight, self.proj.bias, cd)
        return xs, hiddens


def _stream_encoder_step(enc, xs, hiddens):
    """Input LayerNorm + the ResLayerNormLSTM loop (rnnt/models.py:55-75,124,131-134) for a chunk of a few frames in
    bf16, inference only: ``edgedict_stream_encoder_step``.  Returns (out [B, T', H] bf16, (h, c) [L, B, H] fp32)."""
    import ctypes
    lstm = enc.lstm
    B, T, I0 = xs.shape
    H, L = lstm.hidden_size, len(lstm.lstms)
    dev = xs.device
    x = xs.contiguous()
    if x.dtype not in (F32, torch.bfloat16):
        x = x.float()
    if hiddens is None:
        h = torch.zeros(L, B, H, dtype=F32, device=dev)
        c = torch.zeros(L, B, H, dtype=F32, device=dev)
    else:                                                  # new tensors: the caller's state is not modified in place
        h = _state(hiddens[0]).clone()
        c = _state(hiddens[1]).clone()
    cd = torch.bfloat16
    w_ih = [WEIGHTS.get(m.layer(0)[0], cd) for m in lstm.lstms]
    w_hh = [WEIGHTS.get(m.layer(0)[1], cd) for m in lstm.lstms]
    b_ih = [m.layer(0)[2].detach() for m in lstm.lstms]
    b_hh = [m.layer(0)[3].detach() for m in lstm.lstms]
    g = [p[0].weight.detach() for p in lstm.projs]
    bt = [p[0].bias.detach() for p in lstm.projs]
    T_out = T
    for r in lstm.reductions:
        T_out = (T_out + r - 1) // r
    out = torch.empty(B, T_out, H, dtype=cd, device=dev)
    lib = _lib.load()
    ws = torch.empty(lib.edgedict_stream_encoder_workspace_bytes(B, T, I0, H, L), dtype=torch.uint8, device=dev)

    def arr(ts):
        return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    red = (ctypes.c_int * L)(*[int(r) for r in lstm.reductions])
    t_out = ctypes.c_int(0)
    rc = lib.edgedict_stream_encoder_step(
        _lib.ptr(x), _lib.dtype_code(x.dtype), B, T, I0, H, L, _lib.ptr(enc.norm.weight.detach()),
        _lib.ptr(enc.norm.bias.detach()), arr(w_ih), arr(w_hh), arr(b_ih), arr(b_hh), arr(g), arr(bt), red,
        _lib.ptr(h), _lib.ptr(c), _lib.ptr(out), ctypes.byref(t_out), _lib.ptr(ws), _lib.stream_ptr())
    _lib.check(rc, "stream_encoder_step")
    assert t_out.value == T_out
    return out, (h, c)


class Decoder(nn.Module):
    """Prediction network: Embedding -> multi-layer LSTM -> Linear (rnnt/models.py:139-157)."""

    def __init__(self, vocab_embed_size, vocab_size, hidden_size, num_layers,
                 dropout=0, proj_size=None):
        super().__init__()
        self.embed = nn.Embedding(vocab_size, vocab_embed_size, padding_idx=PAD)  # container
        self.lstm = _LSTMParams(vocab_embed_size, hidden_size, num_layers, dropout)
        self.proj = _LinearParams(hidden_size, proj_size)
        self.dropout = dropout

    def forward(self, ys, hidden=None):
        require_cuda(self.embed.weight)
        cd = getattr(self, "compute_dtype", None) or config.get_compute_dtype()
        prepend = hidden is None
        tokens = ys.to(device=self.embed.weight.device, dtype=torch.int32)
        if tokens.dim() != 2:
            raise ValueError("decoder expects token ids of shape [B, U]")
        if not tokens.is_contiguous():
            tokens = tokens.contiguous()
        x = _EmbeddingFn.apply(tokens, self.embed.weight, prepend, cd)
        hs, cs = [], []
        for k in range(self.lstm.num_layers):
            h0 = c0 = None
            if hidden is not None:
                h0, c0 = _state(hidden[0][k]), _state(hidden[1][k])
            w_ih, w_hh, b_ih, b_hh = self.lstm.layer(k)
            x, h, c = _LSTMBlockFn.apply(x, w_ih, w_hh, b_ih, b_hh, None, None, h0, c0,
                                         False, 1, cd)
            # nn.LSTM(dropout=p): on the outputs of every layer except the last, training only
            if self.dropout > 0 and self.training and k + 1 < self.lstm.num_layers:
                x = _dropout(x, self.dropout)
            hs.append(h)
            cs.append(c)
        y = _LinearFn.apply(x, self.proj.weight, self.proj.bias, cd)
        return y, (torch.stack(hs, 0), torch.stack(cs, 0))


class Joint(nn.Module):
    """Linear(P_enc+P_dec, J) -> Tanh -> Linear(J, V) on every (t, u) pair
    (rnnt/models.py:160-179).  Returns raw logits."""

    def __init__(self, input_size, hidden_size, vocab_size):
        super().__init__()
        self.joint = nn.Sequential(_LinearParams(input_size, hidden_size), nn.Tanh(),
                                   _LinearParams(hidden_size, vocab_size))

    def forward(self, h_enc, h_dec):
        require_cuda(h_enc, h_dec)
        cd = getattr(self, "compute_dtype", None) or config.get_compute_dtype()
        l1, l2 = self.joint[0], self.joint[2]
        if h_enc.dim() == 3 and h_dec.dim() == 3:
            e, d = _to_cd(h_enc, cd), _to_cd(h_dec, cd)
            return _JointFn.apply(e, d, l1.weight, l1.bias, l2.weight, l2.bias, cd)
        if h_enc.dim() != h_dec.dim() or h_enc.dim() != 2:
            raise ValueError("joint expects two 3-D or two 2-D inputs")
        e, d = _to_cd(h_enc, cd)[:, None], _to_cd(h_dec, cd)[:, None]
        out = _JointFn.apply(e, d, l1.weight, l1.bias, l2.weight, l2.bias, cd)
        return out[:, 0, 0]


class Transducer(nn.Module):
    """RNN-Transducer with the reference's constructor and methods (rnnt/models.py:182-269)."""

    def __init__(self,
                 vocab_embed_size, vocab_size, input_size,
                 enc_hidden_size, enc_layers, enc_dropout, enc_proj_size,
                 dec_hidden_size, dec_layers, dec_dropout, dec_proj_size,
                 joint_size, enc_time_reductions=[1],
                 blank=NUL, module_type='LSTM', output_loss=True):
        super().__init__()
        self.blank = blank
        if module_type not in ['GRU', 'LSTM']:
            raise ValueError('Unsupported module type')
        self.encoder = Encoder(input_size=input_size, hidden_size=enc_hidden_size,
                               num_layers=enc_layers, dropout=enc_dropout,
                               proj_size=enc_proj_size, time_reductions=enc_time_reductions,
                               module=ResLayerNormGRU if module_type == 'GRU' else ResLayerNormLSTM)
        self.decoder = Decoder(vocab_embed_size=vocab_embed_size, vocab_size=vocab_size,
                               hidden_size=dec_hidden_size, num_layers=dec_layers,
                               dropout=dec_dropout, proj_size=dec_proj_size)
        self.joint = Joint(input_size=enc_proj_size + dec_proj_size, hidden_size=joint_size,
                           vocab_size=vocab_size)
        self.output_loss = output_loss

    # compute dtype shared with the sub-modules -----------------------------------------
    @property
    def compute_dtype(self):
        return getattr(self.encoder, "compute_dtype", None) or config.get_compute_dtype()

    @compute_dtype.setter
    def compute_dtype(self, value):
        value = config._parse(value)
        for m in (self.encoder, self.decoder, self.joint):
            m.compute_dtype = value

    def scale_length(self, logits, xlen):
        # rnnt/models.py:223-226 (host-side integer logic on a [B] tensor)
        scale = (xlen.max().float() / logits.shape[1]).ceil()
        xlen = (xlen / scale).ceil().int()
        return xlen

    def forward(self, xs, ys, xlen, ylen):
        xs = xs[:, :xlen.max()].contiguous()
        ys = ys[:, :ylen.max()].contiguous()
        if xs.is_cuda and not ys.is_cuda:
            # a host-side label batch (seq_collate output with only xs uploaded): one upload here,
            # the prediction network and the loss kernels both read the device copy
            ys = ys.to(xs.device, non_blocking=True)
        if config.DECODER_ON_AUX_STREAM and xs.is_cuda:
            # the prediction network does not depend on the encoder: it runs on the auxiliary
            # stream under the encoder's recurrences.  Autograd replays each node on its forward
            # stream, so its backward overlaps the encoder's backward the same way.
            main = torch.cuda.current_stream(xs.device)
            aux = side.stream(xs.device)
            ready = main.record_event()          # ys (and the parameters) are ready here
            if config.DECODER_ENQUEUE_FIRST:
                aux.wait_event(ready)
                ys.record_stream(aux)
                with torch.cuda.stream(aux):
                    h_dec, _ = self.decoder(ys)
                h_enc, _ = self.encoder(xs)
            else:
                h_enc, _ = self.encoder(xs)          # enqueued first: it is the long pole
                aux.wait_event(ready)
                ys.record_stream(aux)
                with torch.cuda.stream(aux):
                    h_dec, _ = self.decoder(ys)
            main.wait_stream(aux)
            h_dec.record_stream(main)
        else:
            h_enc, _ = self.encoder(xs)
            h_dec, _ = self.decoder(ys)
        ops.mark("joint:enter")
        if (self.output_loss and config.PACKED_LATTICE and not xlen.is_cuda and not ylen.is_cuda
                and h_enc.dim() == 3 and h_enc.is_cuda):
            # lengths on the host: joint + loss on the packed lattice (no padding rows anywhere)
            cd = self.compute_dtype
            l1, l2 = self.joint.joint[0], self.joint.joint[2]
            act = self.scale_length(h_enc, xlen)
            # the loss kernels take a raw device pointer: a host-side label batch (seq_collate
            # output with only xs uploaded) is moved, as Decoder.forward does for its own copy
            labels = ys.to(device=h_enc.device, dtype=torch.int32).contiguous()
            loss = _JointLossFn.apply(_to_cd(h_enc, cd), _to_cd(h_dec, cd), l1.weight, l1.bias,
                                      l2.weight, l2.bias, labels, act, ylen, self.blank, cd)
            ops.mark("joint:exit")
            return loss
        logits = self.joint(h_enc, h_dec)
        ops.mark("joint:exit")
        if self.output_loss:
            xlen = self.scale_length(logits, xlen)
            dev = logits.device   # lengths may live on the host (no sync for the slicing above)
            labels = ys.to(device=dev, dtype=torch.int32).contiguous()
            loss = _RNNTLossFn.apply(
                logits, labels,
                xlen.to(device=dev, dtype=torch.int32, non_blocking=True).contiguous(),
                ylen.to(device=dev, dtype=torch.int32, non_blocking=True).contiguous(),
                self.blank, "mean")
            return loss
        return logits

    @torch.no_grad()
    def greedy_decode(self, xs, xlen):
        """Batched one-symbol-per-frame greedy search (rnnt/models.py:243-269): returns
        (list of int64 numpy arrays INCLUDING blanks, truncated to the un-scaled xlen,
        -sum_t max log p as a [B] tensor)."""
        from .decode import greedy_decode_batch
        return greedy_decode_batch(self, xs, xlen)

    def beam_search(self, xs, xlen=None, W=10, prefix=False, max_expansions=None):
        """Beam search of the reference's legacy model (models.py:121-202), batched; see
        ``decode.beam_search_batch``.  ``prefix=True`` is its prefix-sum variant (:145-161)."""
        from .decode import beam_search_batch
        return beam_search_batch(self, xs, xlen, W, max_expansions, prefix=prefix)


class _CausalConvFn(torch.autograd.Function):
    """Conv1d(C_in, C_out, k, stride s, padding k-1) followed by dropping the last k-1 frames
    (CausalConv1d / DilatedConvBlock of rnnt/models.py:314-339), channels-last: im2col gather +
    one MFMA GEMM; backward = dW GEMM, bias column sum, dcols GEMM + col2im gather."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, cd):
        B, Tin, C = x.shape
        Cout, Cin, k = weight.shape
        assert Cin == C
        cols = ops.conv_im2col(x.contiguous(), k, stride, cd)
        Tout = cols.shape[1]
        w2 = WEIGHTS.get(weight, cd).view(Cout, Cin * k)
        y = ops.gemm(cols.view(B * Tout, C * k), w2, bias=None if bias is None else bias.detach())
        ctx.save_for_backward(cols, weight)
        ctx.cfg = (B, Tin, C, k, stride, cd, bias is not None, x.dtype)
        return y.view(B, Tout, Cout)

    @staticmethod
    def backward(ctx, dy):
        cols, weight = ctx.saved_tensors
        B, Tin, C, k, stride, cd, has_bias, in_dtype = ctx.cfg
        Cout = weight.shape[0]
        Tout = cols.shape[1]
        M = B * Tout
        dy2 = dy.contiguous().view(M, Cout)
        if dy2.dtype != cd:
            dy2 = ops.cast(dy2, cd)
        dw = ops.gemm(dy2.t(), cols.view(M, C * k).t(), out_dtype=F32,
                      split_k=ops.pick_split_k(Cout, C * k, M)).view(Cout, C, k)
        db = ops.colsum(dy2) if has_bias else None
        dx = None
        if ctx.needs_input_grad[0]:
            w2 = WEIGHTS.get(weight, cd).view(Cout, C * k)
            dcols = ops.gemm(dy2, w2.t()).view(B, Tout, C * k)
            dx = ops.conv_col2im(dcols, Tin, C, k, stride)
            if dx.dtype != in_dtype:
                dx = ops.cast(dx, in_dtype)
        return dx, dw, db, None, None


class _GeluGroupNormFn(torch.autograd.Function):
    """GroupNorm(1, C)(GELU(y)) on channels-last [B,T,C] (DilatedConvBlock.forward,
    rnnt/models.py:333-335): statistics over all (t, c) of a sample, affine per channel."""

    @staticmethod
    def forward(ctx, y, gamma, beta):
        out, mean, rstd = ops.gelu_groupnorm_fwd(y.contiguous(), gamma.detach(), beta.detach())
        ctx.save_for_backward(y, gamma, mean, rstd)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, gamma, mean, rstd = ctx.saved_tensors
        if dout.dtype != y.dtype:
            dout = ops.cast(dout.contiguous(), y.dtype)
        dy, dgamma, dbeta = ops.gelu_groupnorm_bwd(y, dout, gamma.detach(), mean, rstd)
        return dy, dgamma, dbeta


class _ConvParams(nn.Module):
    """nn.Conv1d-compatible parameters: weight [C_out, C_in, k] (kaiming-normal as the reference
    re-initialises it, rnnt/models.py:317,329), bias [C_out] (nn.Conv1d default init)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = (kernel_size,), (stride,)
        self.padding = (kernel_size - 1,)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size))
        nn.init.kaiming_normal_(self.weight)
        if bias:
            bound = 1.0 / math.sqrt(in_channels * kernel_size)
            self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)


class DilatedConvBlock(nn.Module):
    """GELU -> GroupNorm(group_norm_size=1, C_in) -> causal strided Conv1d (rnnt/models.py:323-339);
    channels-last in and out.  Only dilation 1 and one group (what FrontEnd builds)."""

    def __init__(self, in_channels, out_channels, kernel_size, dilation=1, group_norm_size=1,
                 stride=1, bias=True):
        super().__init__()
        if dilation != 1 or group_norm_size != 1:
            raise NotImplementedError("DilatedConvBlock: only dilation=1, group_norm_size=1 (FrontEnd's use)")
        self.conv = _ConvParams(in_channels, out_channels, kernel_size, stride, bias)
        self.gn = _LayerNormParams(in_channels)      # weight / bias [C_in], the GroupNorm affine

    def forward(self, x, cd):
        x = _GeluGroupNormFn.apply(x, self.gn.weight, self.gn.bias)
        return _CausalConvFn.apply(x, self.conv.weight, self.conv.bias, self.conv.stride[0], cd)


class FrontEnd(nn.Module):
    """Convolutional waveform feature extractor of the reference's cli/train.py
    (rnnt/models.py:341-365): f32 [B, N] (or [B,1,N]) -> [B, T, C_last], LayerNorm over channels.
    State-dict keys as the reference: conv1.{weight,bias}, encode.{i}.conv.{weight,bias},
    encode.{i}.gn.{weight,bias}, layer_norm.{weight,bias}."""

    def __init__(self, frontend_params=[(10, 5, 16)] + [(8, 4, 32)] + [(4, 2, 128)] * 3, bias=True):
        super().__init__()
        ks = [p[0] for p in frontend_params]
        st = [p[1] for p in frontend_params]
        ch = [p[2] for p in frontend_params]
        self.conv1 = _ConvParams(1, ch[0], ks[0], st[0], bias)
        self.encode = nn.Sequential(*[
            DilatedConvBlock(ch[i - 1], ch[i], ks[i], dilation=1, stride=st[i], group_norm_size=1, bias=bias)
            for i in range(1, len(st))])
        self.layer_norm = _LayerNormParams(ch[-1])

    def out_frames(self, n_samples):
        t = ops.conv_out_frames(n_samples, self.conv1.kernel_size[0], self.conv1.stride[0])
        for blk in self.encode:
            t = ops.conv_out_frames(t, blk.conv.kernel_size[0], blk.conv.stride[0])
        return t

    def forward(self, x):
        require_cuda(x)
        cd = getattr(self, "compute_dtype", None) or config.get_compute_dtype()
        if not isinstance(cd, torch.dtype):
            cd = config._parse(cd)
        if x.dim() == 3:                      # B x 1 x T -> B x T (rnnt/models.py:352-353 the other way)
            x = x[:, 0]
        x = x.float().contiguous().unsqueeze(-1)          # channels-last [B, N, 1]
        x = _CausalConvFn.apply(x, self.conv1.weight, self.conv1.bias, self.conv1.stride[0], cd)
        for blk in self.encode:
            x = blk(x, cd)
        return _InputNormFn.apply(x, self.layer_norm.weight, self.layer_norm.bias, cd)


class _OutsideHotPath(nn.Module):
    """Names the reference's scripts import next to ``Transducer`` (cli/train.py:18,
    rnnt/wav2vec.py:12) but whose arithmetic is not on the MI355X hot path (SURVEY.md 8f rank 4,
    DESIGN.md section 8).  They stay importable so that those scripts load; constructing one fails
    loudly - there is no eager-PyTorch fallback in this engine."""

    _what = ""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError(
            "%s is not implemented by the MI355X engine (%s); use the log-mel front-end and the LSTM "
            "encoder (cli/baseline.py path)" % (type(self).__name__, self._what))


class CTCEncoder(_OutsideHotPath):
    _what = "the CTC encoder of rnnt/models.py:272-311"


def convert_lightning2normal(checkpoint):
    """Lightning checkpoint -> ``{'model': state_dict}`` (same contract as the reference's
    rnnt/models.py:366-380): when the file has a ``state_dict`` whose keys carry the Lightning
    ``model.`` prefix, the prefix is removed everywhere and the result is wrapped under
    ``'model'``; a ``state_dict`` without the prefix is returned bare; anything else is
    passed through untouched."""
    if 'state_dict' not in checkpoint:
        return checkpoint
    inner = checkpoint['state_dict']
    names = list(inner.keys())
    if names and 'model.' in names[0]:
        return {'model': {name.replace('model.', ''): t for name, t in inner.items()}}
    return inner
