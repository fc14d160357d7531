"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement of the reference's transducer beam search, the legacy root-level
``Transducer.beam_search`` (/root/reference/models.py:121-202, with ``Sequence`` :212-224, including
the ``prefix=True`` branch :145-161), driven by the MAINTAINED model's arithmetic
(oracle/models_ref.py: rnnt/models.py encoder, prediction network, joint).  Graves (2012) search,
one utterance at a time, as written there:

    B = [empty hypothesis, logp 0]
    for each encoder frame x:
        A = B; B = []
        loop:
            y* = the most probable hypothesis in A (first one on ties: Python ``max``); remove it
            pred, hidden = prediction network on y*'s LAST token from y*'s stored state
            logp = log_softmax(joint(x, pred))
            for every k in the vocabulary: y* + k with logp(y*) + logp[k]  (Python float = fp64 sum
                of fp32 log-probabilities); k == blank goes to B and keeps y*'s tokens AND y*'s
                stored state; the others go to A in k order with state = hidden
            stop when len(B) >= W and max(B) >= max(A)
        B = B[:W]        # the ``sorted(...)`` calls at :141,:195 discard their result, so this keeps
                         # the first W hypotheses in INSERTION order, as the reference does
    return B[0].tokens, -B[0].logp          # likewise the first inserted one, not the best

``prefix=True`` (models.py:145-161) adds, at the start of every frame and before anything is expanded, the
probability that a hypothesis A[j] is reached THROUGH a hypothesis A[i] that is a proper prefix of it and comes
LATER in the list (i > j): starting from logp(A[i]), the remaining tokens of A[j] are emitted on THIS frame - each
with the prediction-network output ``g`` stored when that position was expanded (the first one is recomputed from
A[i]'s last token and state, which gives the same vector) - and the sum is folded into logp(A[j]) with
``log_aplusb``; pairs are visited j ascending, i ascending, and A[i]'s own logp is read before its turn as a "j".

Adaptations to the maintained model (the legacy class has its own layers and feeds token id 1 from
a ``None`` state first): the empty hypothesis' "last token" is BOS with zero state - exactly what
``Decoder.forward(empty, None)`` does (rnnt/models.py:150-153) and what ``greedy_decode`` starts
from (:247); returned tokens exclude it.  Frames: an utterance of a batch uses the first
``scale_length(T', xlen)`` encoder frames (rnnt/models.py:223-226); batch-1 with a full-length
``xlen`` is the reference's loop over every frame.

PARITY STATUS: **pinned on the reference's own code, executed** (round 3).  oracle/make_golden_beam.py lifts
``Transducer.beam_search``, ``Sequence`` and ``log_aplusb`` out of /root/reference/models.py:121-224 and
runs them unmodified on a stub ``self`` backed by the reference's maintained ``rnnt.models`` sub-modules
(on torch 2.10 ``autograd.Variable(..., volatile=True)`` merely warns; only the module FILE cannot be
imported, for its absent ``recurrent`` dependency); for W = 1, 2, 4, 10 on the trained tiny model this
restatement returns the same tokens and the same number of hypothesis expansions, and scores equal to 1e-6
(``ref_*`` arrays of tests/golden/beam_tiny.npz; tests/test_oracle_beam.py).  The reference holds no
fixture of its own for this method.
"""
import numpy as np
import torch

from . import models_ref as M


class _Hyp:
    __slots__ = ("k", "tok", "h", "logp", "g")

    def __init__(self, k, tok, h, logp, g=None):
        self.k, self.tok, self.h, self.logp = k, tok, h, logp
        self.g = g if g is not None else []     # prefix=True: prediction-network outputs, g[m] predicts k[m]


def log_aplusb(a, b):
    import math
    return max(a, b) + math.log1p(math.exp(-math.fabs(a - b)))


def _isprefix(a, b):
    if a == b or len(a) >= len(b):
        return False
    return all(a[i] == b[i] for i in range(len(a)))


def beam_search_one(sd, h_enc, W=10, blank=M.NUL, prefix=False):
    """h_enc [T, P_enc] fp32 of ONE utterance -> (token list without blanks, -logp, expansions).
    (k holds the tokens WITHOUT the legacy start token, so the reference's k[m + 1] is k[m] here and its g[m] - the
    prediction after consuming its k[m] - is g[m] here, the prediction that emits k[m].)"""
    L = M.n_dec_layers(sd)
    H = sd["decoder.lstm.weight_hh_l0"].shape[1]
    zero = (torch.zeros(L, 1, H), torch.zeros(L, 1, H))
    V = sd["joint.joint.2.weight"].shape[0]
    B = [_Hyp([], M.BOS, zero, 0.0)]
    n_expansions = 0
    for x in h_enc:
        A = B
        B = []
        if prefix:
            for j in range(len(A) - 1):
                for i in range(j + 1, len(A)):
                    if not _isprefix(A[i].k, A[j].k):
                        continue
                    pred, _ = M.decoder_forward(sd, torch.tensor([[A[i].tok]]), A[i].h)
                    n_expansions += 1           # (counts prediction-network steps, as the executed reference's stub does)
                    idx = len(A[i].k)
                    logp = torch.log_softmax(M.joint_forward(sd, x[None, :], pred[:, 0])[0], dim=0)
                    cur = A[i].logp + float(logp[A[j].k[idx]])
                    for m in range(idx + 1, len(A[j].k)):
                        logp = torch.log_softmax(M.joint_forward(sd, x[None, :], A[j].g[m][None, :])[0], dim=0)
                        cur += float(logp[A[j].k[m]])
                    A[j].logp = log_aplusb(A[j].logp, cur)
        while True:
            y_hat = max(A, key=lambda a: a.logp)
            A.remove(y_hat)
            pred, hidden = M.decoder_forward(sd, torch.tensor([[y_hat.tok]]), y_hat.h)
            logits = M.joint_forward(sd, x[None, :], pred[:, 0])[0]
            logp = torch.log_softmax(logits, dim=0)
            n_expansions += 1
            for k in range(V):
                lp = y_hat.logp + float(logp[k])
                if k == blank:
                    B.append(_Hyp(y_hat.k, y_hat.tok, y_hat.h, lp, y_hat.g))
                else:
                    A.append(_Hyp(y_hat.k + [k], k, hidden, lp, (y_hat.g + [pred[0, 0]]) if prefix else None))
            y_a = max(A, key=lambda a: a.logp)
            y_b = max(B, key=lambda a: a.logp)
            if len(B) >= W and y_b.logp >= y_a.logp:
                break
        B = B[:W]
    return list(B[0].k), -B[0].logp, n_expansions


def beam_search(sd, xs, xlen=None, W=10, blank=M.NUL, time_reductions=(1,), prefix=False):
    """xs [B, T0, I]; returns (list of int64 arrays, fp64 scores [B], total expansions)."""
    h_enc, _ = M.encoder_forward(sd, xs, None, time_reductions)
    Bn, T = h_enc.shape[0], h_enc.shape[1]
    lens = [T] * Bn if xlen is None else [int(v) for v in M.scale_length(T, xlen)]
    seqs, scores, total = [], [], 0
    for b in range(Bn):
        k, s, n = beam_search_one(sd, h_enc[b, :lens[b]], W, blank, prefix)
        seqs.append(np.array(k, dtype=np.int64))
        scores.append(s)
        total += n
    return seqs, np.array(scores, dtype=np.float64), total
