"""CPU: the beam-search oracle (oracle/beam_ref.py) against the vectors the REFERENCE's own legacy search
produced when executed (``ref_*`` arrays of tests/golden/beam_tiny.npz: oracle/make_golden_beam.py lifts
``Transducer.beam_search`` / ``Sequence`` / ``log_aplusb`` out of /root/reference/models.py:121-224 and runs
them on the reference's maintained sub-modules), and structural properties of the search."""
import os

import numpy as np
import pytest
import torch

from oracle import beam_ref, models_ref as M

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "beam_tiny.npz"))
WIDTHS = (1, 2, 4, 10)


def load():
    sd = {k[3:]: torch.from_numpy(G[k]) for k in G.files if k.startswith("sd/")}
    return sd, torch.from_numpy(G["xs"]), torch.from_numpy(G["xlen"])


@pytest.mark.parametrize("W", WIDTHS)
def test_oracle_reproduces_committed_vectors(W):
    sd, xs, xlen = load()
    seqs, scores, n = beam_ref.beam_search(sd, xs, xlen, W=W)
    for b, s in enumerate(seqs):
        assert np.array_equal(s, G["W%d_seq%d" % (W, b)])
    np.testing.assert_allclose(scores, G["W%d_score" % W], rtol=1e-6)
    assert n == int(G["W%d_expansions" % W][0])


@pytest.mark.parametrize("W", WIDTHS)
def test_oracle_reproduces_the_reference_executed_search(W):
    """tokens and the number of hypothesis expansions exactly, scores to 1e-6 (fp32 module arithmetic of the
    reference vs the oracle's functional restatement, summed in fp64 on both sides)."""
    sd, xs, xlen = load()
    seqs, scores, n = beam_ref.beam_search(sd, xs, xlen, W=W)
    for b, s in enumerate(seqs):
        assert np.array_equal(s, G["ref_W%d_seq%d" % (W, b)])
    np.testing.assert_allclose(scores, G["ref_W%d_score" % W], rtol=1e-6)
    assert n == int(G["ref_W%d_expansions" % W][0])


@pytest.mark.parametrize("W", WIDTHS[1:])
def test_oracle_prefix_branch_reproduces_the_reference_executed_search(W):
    """``prefix=True`` (models.py:145-161), the lifted reference function run with the flag set: tokens and the number
    of prediction-network steps exactly, scores to 1e-6; and the branch is not a no-op on these vectors (the scores
    move by 0.15 - 1.1 against prefix=False)."""
    sd, xs, xlen = load()
    seqs, scores, n = beam_ref.beam_search(sd, xs, xlen, W=W, prefix=True)
    for b, s in enumerate(seqs):
        assert np.array_equal(s, G["ref_P_W%d_seq%d" % (W, b)])
        assert np.array_equal(s, G["P_W%d_seq%d" % (W, b)])
    np.testing.assert_allclose(scores, G["ref_P_W%d_score" % W], rtol=1e-6)
    assert n == int(G["ref_P_W%d_expansions" % W][0]) == int(G["P_W%d_expansions" % W][0])
    assert np.abs(G["ref_P_W%d_score" % W] - G["ref_W%d_score" % W]).max() > 0.1
    assert n > int(G["ref_W%d_expansions" % W][0])


def test_score_is_the_log_probability_of_one_alignment():
    """-score must be <= the total log-probability of the returned label sequence (it is the
    probability of a single path through the lattice), and for W = 10 the trained model recovers
    two of the three training transcripts."""
    from oracle import rnnt_loss_ref as R
    sd, xs, xlen = load()
    seqs, scores, _ = beam_ref.beam_search(sd, xs, xlen, W=10)
    ys, ylen = G["ys"], G["ylen"]
    assert sum(np.array_equal(s, ys[b, :ylen[b]]) for b, s in enumerate(seqs)) >= 2
    h_enc, _ = M.encoder_forward(sd, xs)
    lens = M.scale_length(h_enc.shape[1], xlen)
    for b, s in enumerate(seqs):
        lab = torch.from_numpy(s)[None].int()
        h_dec, _ = M.decoder_forward(sd, lab, None)
        logits = M.joint_forward(sd, h_enc[b:b + 1, :int(lens[b])], h_dec)
        cost, _ = R.rnnt_loss(logits.double().numpy(), lab.numpy(), np.array([int(lens[b])], dtype=np.int32),
                              np.array([len(s)], dtype=np.int32), want_grads=False)
        assert scores[b] >= cost[0] - 1e-4       # one path <= sum over paths


def test_every_frame_needs_at_least_W_expansions():
    sd, xs, xlen = load()
    frames = int(M.scale_length((xs.shape[1] + 1) // 2, xlen).sum())
    for W in WIDTHS:
        assert int(G["W%d_expansions" % W][0]) >= W * frames
