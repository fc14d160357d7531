"""ORACLE of the ORACLE (test infrastructure only - never imported by the product path).

The RNN-T loss straight from its DEFINITION (Graves 2012, eq. 1-4 / SURVEY.md appendix A7), with no
alpha/beta recursion at all: for one utterance with T frames and labels y_1..y_U

    P(y | x) = sum over every monotone alignment a of  prod_i p(a_i | t_i, u_i)

where an alignment is an interleaving of exactly T blanks and the U labels (in order) that ENDS with a
blank: a blank at lattice node (t, u) moves to (t+1, u), label y_{u+1} at (t, u) moves to (t, u+1), the
walk starts at (0, 0) and the last blank is emitted at (T-1, U).  There are C(T+U-1, U) such walks; they
are enumerated explicitly, each one's log-probability is the sum of its log-softmaxed logits, and

    cost = -log sum_a exp(logp_a)

is formed in float64 torch, so autograd through the enumeration gives the gradient w.r.t. the raw logits.
This shares nothing with oracle/rnnt_loss_ref.py's dynamic programme except the log-softmax, so it does not
share the recursion's failure modes (off-by-one at the lattice borders, the final-blank term, the label
index of a cell, lengths vs tensor extents).  Only usable on tiny lattices (T <= 5, U <= 4).

Why this exists: the reference takes its loss from the third-party, un-vendored warprnnt_pytorch
(rnnt/models.py:221,238) and holds no fixture for it ("parity unpinned by the reference"); the DP oracle was
anchored on one upstream known-answer vector only.
"""
import itertools

import numpy as np
import torch


def alignments(T, U):
    """Every walk as a tuple of booleans (True = emit the next label), length T + U - 1; the final blank at
    (T-1, U) is implicit."""
    n = T + U - 1
    for pos in itertools.combinations(range(n), U):
        s = [False] * n
        for p in pos:
            s[p] = True
        yield tuple(s)


def cost_one(logits, labels, T, U, blank=0):
    """logits: float64 tensor [>=T, >=U+1, V] (raw); labels: sequence of >= U ids.  Returns the cost
    (0-dim float64 tensor, differentiable w.r.t. ``logits``) and the number of alignments."""
    lp = torch.log_softmax(logits[:T, :U + 1].double(), dim=-1)
    terms = []
    for walk in alignments(T, U):
        t = u = 0
        acc = lp.new_zeros(())
        for emit in walk:
            if emit:
                acc = acc + lp[t, u, int(labels[u])]
                u += 1
            else:
                acc = acc + lp[t, u, blank]
                t += 1
        assert t == T - 1 and u == U, (t, u)
        terms.append(acc + lp[T - 1, U, blank])
    return -torch.logsumexp(torch.stack(terms), 0), len(terms)


def rnnt_loss(acts, labels, act_lens, label_lens, blank=0):
    """acts [B, T, U1, V] (numpy) -> (costs [B] float64, grads [B, T, U1, V] float64 of sum_b cost_b)."""
    a = torch.tensor(np.asarray(acts, dtype=np.float64), requires_grad=True)
    costs = []
    for b in range(a.shape[0]):
        c, _ = cost_one(a[b], labels[b], int(act_lens[b]), int(label_lens[b]), blank)
        costs.append(c)
    total = torch.stack(costs)
    total.sum().backward()
    return total.detach().numpy(), a.grad.numpy()
