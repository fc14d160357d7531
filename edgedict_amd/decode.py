"""Greedy search over a batch, driven by csrc/decode.hip (one C call for the whole time loop).

``greedy_decode_batch`` reproduces ``Transducer.greedy_decode`` (rnnt/models.py:243-269):
encoder over the whole batch, prediction network primed with BOS, then for each encoder frame
joint -> log-softmax max -> decoder step for every row -> state committed only where the
symbol is not blank; blanks stay in the returned sequences, which are truncated to the
(un-scaled) ``xlen``; the score is ``-sum_t max log p``.
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
    lens = xlen.cpu().numpy() if torch.is_tensor(xlen) else np.asarray(xlen)
    return [seq[:int(n)] for seq, n in zip(toks, lens)], score
