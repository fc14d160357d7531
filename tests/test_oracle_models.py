"""CPU: pin the model oracle on the golden outputs recorded from the reference module
(oracle/make_golden.py ran /root/reference's rnnt.models.Transducer in the build container)."""
import os

import numpy as np
import pytest
import torch

from oracle import models_ref as M
from oracle import rnnt_loss_ref as R
from oracle.make_golden import CASES

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    cfg, B, T0, U, seed = CASES[name]
    g = np.load(os.path.join(GOLD, "transducer_%s.npz" % name))
    sd = M.make_state_dict(cfg, seed)
    batch = M.make_batch(cfg, seed + 1, B, T0, U)
    return cfg, sd, batch, g


@pytest.mark.parametrize("name", ["tiny", "gru_tiny", "E4D1", "E6D2", "E6D2_LARGE"])
def test_oracle_reproduces_reference_outputs(name):
    cfg, sd, (xs, ys, xlen, ylen), g = _load(name)
    assert np.array_equal(xlen.numpy(), g["xlen"]) and np.array_equal(ylen.numpy(), g["ylen"])
    with torch.no_grad():
        logits, act_lens = M.transducer_logits(sd, xs, ys, xlen, ylen)
    assert np.array_equal(act_lens.numpy(), g["act_lens"])
    if "logits" in g.files:
        np.testing.assert_allclose(logits.numpy(), g["logits"], atol=2e-5)
    else:
        np.testing.assert_allclose(logits.numpy()[:, ::7, ::3, ::64], g["logits_sample"], atol=2e-5)
    costs, _ = R.rnnt_loss(logits.double().numpy(), ys.numpy(), act_lens.numpy(), ylen.numpy(),
                           want_grads=False)
    np.testing.assert_allclose(costs, g["costs"], rtol=1e-6)


@pytest.mark.parametrize("name", ["tiny", "gru_tiny", "E4D1", "E6D2"])
def test_oracle_greedy_tokens_bit_exact(name):
    cfg, sd, (xs, ys, xlen, ylen), g = _load(name)
    with torch.no_grad():
        tokens, score = M.greedy_decode(sd, xs, xlen)
    for b, t in enumerate(tokens):
        assert np.array_equal(t, g["greedy_tokens"][b][:len(t)])
    np.testing.assert_allclose(score.numpy(), g["greedy_score"], rtol=1e-5)


def test_explicit_recurrence_matches_library_lstm():
    cfg, sd, (xs, ys, xlen, ylen), g = _load("tiny")
    with torch.no_grad():
        a, _ = M.transducer_logits(sd, xs, ys, xlen, ylen, explicit=False)
        b, _ = M.transducer_logits(sd, xs, ys, xlen, ylen, explicit=True)
    assert (a - b).abs().max().item() < 1e-5


def test_time_reduction_odd_tail_is_halved():
    x = torch.ones(1, 3, 2)
    y = M.time_reduction(x)
    assert y.shape == (1, 2, 2)
    assert torch.equal(y[0, :, 0], torch.tensor([1.0, 0.5]))


def test_scale_length_matches_survey_probe():
    out = M.scale_length(84, torch.tensor([167, 150, 120, 100], dtype=torch.int32))
    assert out.tolist() == [84, 75, 60, 50]


def test_encoder_chunked_equals_full_when_chunks_even():
    cfg, sd, (xs, ys, xlen, ylen), g = _load("tiny")
    xs = xs[:, :10]
    with torch.no_grad():
        full, _ = M.encoder_forward(sd, xs)
        hid = None
        parts = []
        for s in range(0, 10, 2):
            y, hid = M.encoder_forward(sd, xs[:, s:s + 2], hid)
            parts.append(y)
    assert (torch.cat(parts, 1) - full).abs().max().item() < 1e-5
