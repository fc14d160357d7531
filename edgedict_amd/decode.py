"""Greedy search over a batch, driven by csrc/decode.hip (one C call for the whole time loop).

``greedy_decode_batch`` reproduces ``Transducer.greedy_decode`` (rnnt/models.py:243-269):
encoder over the whole batch, prediction network primed with BOS, then for each encoder frame
joint -> log-softmax max -> decoder step for every row -> state committed only where the
symbol is not blank; blanks stay in the returned sequences, which are truncated to the
(un-scaled) ``xlen``; the score is ``-sum_t max log p``.

``beam_search_batch`` is the reference's legacy ``Transducer.beam_search`` (models.py:121-202,
``prefix=False``) for a batch of utterances in lockstep (csrc/decode.hip, second half).
"""
import ctypes

import numpy as np
import torch

from . import _lib, ops
from ._lib import dtype_code


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class SearchState:
    """Prediction-network state of a batch of hypotheses (one per row)."""

    def __init__(self, dec_out, h, c):
        self.dec_out = dec_out      # [B, P_dec] compute dtype
        self.h = h                  # [L, B, H] fp32
        self.c = c


def init_search_state(model, batch):
    """decoder(BOS) for every row — rnnt/models.py:247, rnnt/stream.py:84-91."""
    dev = model.decoder.embed.weight.device
    empty = torch.empty(batch, 0, dtype=torch.int32, device=dev)
    dec, (h, c) = model.decoder(empty)
    return SearchState(dec[:, 0].contiguous(), h.contiguous(), c.contiguous())


def run_search(model, enc_out, state, unk=-1, want_score=True):
    """Advance ``state`` over all frames of ``enc_out`` [B, T, P_enc] (compute dtype).
    Returns (tokens int32 [B, T] on device, score fp32 [B] or None)."""
    from .models import WEIGHTS
    cd = enc_out.dtype
    B, T, P = enc_out.shape
    dec = model.decoder
    l1, l2 = model.joint.joint[0], model.joint.joint[2]
    J, V = l1.weight.shape[0], l2.weight.shape[0]
    P2 = dec.proj.weight.shape[0]
    L, H = dec.lstm.num_layers, dec.lstm.hidden_size
    E = dec.embed.weight.shape[1]
    w1c = WEIGHTS.get(l1.weight, cd)
    w2c = WEIGHTS.get(l2.weight, cd)
    wpc = WEIGHTS.get(dec.proj.weight, cd)
    # encoder half of the joint's first Linear for all frames at once
    E1 = ops.gemm(enc_out.reshape(B * T, P), w1c[:, :P]) if T > 0 else enc_out.new_empty(0, J)
    w_ih = [WEIGHTS.get(dec.lstm.layer(k)[0], cd) for k in range(L)]
    w_hh = [WEIGHTS.get(dec.lstm.layer(k)[1], cd) for k in range(L)]
    b_ih = [dec.lstm.layer(k)[2].detach() for k in range(L)]
    b_hh = [dec.lstm.layer(k)[3].detach() for k in range(L)]
    lib = _lib.load()
    dev = enc_out.device
    nbytes = lib.edgedict_greedy_workspace_bytes(dtype_code(cd), B, J, V, E, L, H, P2)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    tokens = torch.empty(B, max(T, 1), dtype=torch.int32, device=dev)
    score = torch.zeros(B, dtype=torch.float32, device=dev) if want_score else None
    w1d = w1c[:, P:]
    keep = (E1, w1c, w2c, wpc, w_ih, w_hh, b_ih, b_hh, ws)   # alive until the stream is done
    rc = lib.edgedict_greedy_decode(
        dtype_code(cd), _lib.ptr(E1), ctypes.c_longlong(T * J), ctypes.c_longlong(J), B, T, J,
        _lib.ptr(w1d), ctypes.c_longlong(w1c.stride(0)), _lib.ptr(l1.bias.detach()), P2,
        _lib.ptr(w2c), _lib.ptr(l2.bias.detach()), V, _lib.ptr(dec.embed.weight.detach()),
        dtype_code(dec.embed.weight.dtype), E, L, _ptr_array(w_ih), _ptr_array(w_hh),
        _ptr_array(b_ih), _ptr_array(b_hh), H, _lib.ptr(wpc), _lib.ptr(dec.proj.bias.detach()),
        _lib.ptr(state.h), _lib.ptr(state.c), _lib.ptr(state.dec_out), int(model.blank), int(unk),
        _lib.ptr(tokens), tokens.stride(0), _lib.ptr(score), _lib.ptr(ws), _lib.stream_ptr())
    _lib.check(rc, "greedy_decode")
    del keep
    return tokens[:, :T], score


def greedy_decode_batch(model, xs, xlen):
    _lib.require_cuda(xs)
    enc_out, _ = model.encoder(xs)
    state = init_search_state(model, xs.shape[0])
    tokens, score = run_search(model, enc_out.contiguous(), state, unk=-1, want_score=True)
    toks = tokens.cpu().numpy().astype(np.int64)
    # the copy above synchronised: a bounded in-kernel wait of the encoder stack that gave up during THIS call
    # has written its code by now (an inference call has no later step that would notice it)
    from . import encoder_stack
    encoder_stack.check_wsr_error()
    lens = xlen.cpu().numpy() if torch.is_tensor(xlen) else np.asarray(xlen)
    return [seq[:int(n)] for seq, n in zip(toks, lens)], score


def beam_search_batch(model, xs, xlen=None, W=10, max_expansions=None, prefix=False):
    """Graves (2012) beam search as the reference's legacy ``Transducer.beam_search`` runs it
    (models.py:121-202), batched: every utterance keeps its own A / B sets and all open utterances
    advance one expansion per lockstep iteration on the device.  ``prefix=True`` is the reference's
    prefix-sum variant (:145-161): see ``edgedict_beam_search`` in include/edgedict_hip.h.

    xs [B, T0, I]; xlen (host or device int tensor, stacked frames) or None for "all frames".
    Returns ``(list of int64 arrays (tokens, no blanks), fp64 tensor [B] = -log p)``: per
    utterance the FIRST hypothesis of the last frame's B list, which is what the reference returns
    (its ``sorted`` calls are no-ops).  ``max_expansions`` bounds the pops per utterance and frame
    (default 8 W, at least 16); hitting it raises instead of truncating the search."""
    from .models import WEIGHTS
    _lib.require_cuda(xs)
    if W < 1:
        raise ValueError("beam width must be >= 1")
    enc_out, _ = model.encoder(xs)
    enc_out = enc_out.contiguous()
    cd = enc_out.dtype
    B, T, P = enc_out.shape
    if xlen is None:
        lens = np.full(B, T, dtype=np.int32)
    else:
        xl = xlen.detach().cpu() if torch.is_tensor(xlen) else torch.as_tensor(xlen)
        lens = model.scale_length(enc_out, xl).numpy().astype(np.int32)
    EM = int(max_expansions) if max_expansions else max(16, 8 * W)
    dec = model.decoder
    l1, l2 = model.joint.joint[0], model.joint.joint[2]
    J, V = l1.weight.shape[0], l2.weight.shape[0]
    P2 = dec.proj.weight.shape[0]
    L, H = dec.lstm.num_layers, dec.lstm.hidden_size
    E = dec.embed.weight.shape[1]
    w1c = WEIGHTS.get(l1.weight, cd)
    w2c = WEIGHTS.get(l2.weight, cd)
    wpc = WEIGHTS.get(dec.proj.weight, cd)
    E1 = ops.gemm(enc_out.reshape(B * T, P), w1c[:, :P]) if T > 0 else enc_out.new_empty(0, J)
    w_ih = [WEIGHTS.get(dec.lstm.layer(k)[0], cd) for k in range(L)]
    w_hh = [WEIGHTS.get(dec.lstm.layer(k)[1], cd) for k in range(L)]
    b_ih = [dec.lstm.layer(k)[2].detach() for k in range(L)]
    b_hh = [dec.lstm.layer(k)[3].detach() for k in range(L)]
    lib = _lib.load()
    nbytes = lib.edgedict_beam_workspace_bytes(dtype_code(cd), B, T, J, V, E, L, H, P2, W, EM, int(bool(prefix)))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=enc_out.device)
    max_tokens = T * EM + 1
    tokens = np.zeros((B, max_tokens), dtype=np.int32)
    ntok = np.zeros(B, dtype=np.int32)
    score = np.zeros(B, dtype=np.float64)
    nexp = ctypes.c_longlong(0)
    w1d = w1c[:, P:]
    from .tokenizer import BOS
    rc = lib.edgedict_beam_search(
        dtype_code(cd), _lib.ptr(E1), ctypes.c_longlong(T * J), ctypes.c_longlong(J), B, T,
        lens.ctypes.data_as(ctypes.c_void_p), J, _lib.ptr(w1d), ctypes.c_longlong(w1c.stride(0)),
        _lib.ptr(l1.bias.detach()), P2, _lib.ptr(w2c), _lib.ptr(l2.bias.detach()), V,
        _lib.ptr(dec.embed.weight.detach()), dtype_code(dec.embed.weight.dtype), E, L,
        _ptr_array(w_ih), _ptr_array(w_hh), _ptr_array(b_ih), _ptr_array(b_hh), H, _lib.ptr(wpc),
        _lib.ptr(dec.proj.bias.detach()), int(model.blank), int(BOS), int(W), EM, int(bool(prefix)),
        tokens.ctypes.data_as(ctypes.c_void_p), max_tokens, ntok.ctypes.data_as(ctypes.c_void_p),
        score.ctypes.data_as(ctypes.c_void_p), ctypes.byref(nexp), _lib.ptr(ws), _lib.stream_ptr())
    _lib.check(rc, "beam_search")
    beam_search_batch.last_expansions = int(nexp.value)
    seqs = [tokens[b, :ntok[b]].astype(np.int64) for b in range(B)]
    return seqs, torch.from_numpy(score)
