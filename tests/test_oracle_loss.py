"""CPU: pin the RNN-T loss oracle on the upstream known-answer vector and on autograd."""
import numpy as np
import torch

from oracle import rnnt_loss_ref as R


def _random_case(seed, B=3, T=7, U1=5, V=11):
    rng = np.random.default_rng(seed)
    acts = rng.normal(size=(B, T, U1, V))
    labels = rng.integers(1, V, size=(B, U1 - 1)).astype(np.int32)
    act_lens = rng.integers(1, T + 1, size=B).astype(np.int32)
    label_lens = rng.integers(0, U1, size=B).astype(np.int32)
    act_lens[0] = T
    label_lens[0] = U1 - 1
    return acts, labels, act_lens, label_lens


def test_known_answer_cost_and_grad():
    ka = R.KNOWN_ANSWER
    costs, grads = R.rnnt_loss(ka["acts"], ka["labels"], ka["act_lens"], ka["label_lens"])
    assert abs(costs[0] - ka["cost"]) < 2e-6
    np.testing.assert_allclose(grads, ka["grads"], atol=5e-7)


def test_alpha_beta_agree():
    acts, labels, al, ll = _random_case(1)
    for b in range(acts.shape[0]):
        T, U = int(al[b]), int(ll[b])
        lp = R.log_softmax(acts[b, :T, :U + 1])
        alpha, beta, loglike = R.lattice(lp, labels[b], T, U)
        assert abs(beta[0, 0] - loglike) < 1e-12


def test_analytic_grad_matches_autograd():
    for seed in range(3):
        acts, labels, al, ll = _random_case(seed)
        costs, grads = R.rnnt_loss(acts, labels, al, ll)
        ta = torch.tensor(acts, requires_grad=True)
        ct = R.rnnt_loss_torch(ta, torch.tensor(labels), torch.tensor(al), torch.tensor(ll))
        ct.sum().backward()
        np.testing.assert_allclose(ct.detach().numpy(), costs, rtol=0, atol=1e-12)
        np.testing.assert_allclose(ta.grad.numpy(), grads, rtol=0, atol=1e-12)


def test_vectorised_port_matches_loop_oracle():
    for seed in range(3):
        acts, labels, al, ll = _random_case(10 + seed, B=4, T=9, U1=6, V=13)
        costs, grads = R.rnnt_loss(acts, labels, al, ll)
        cf, gf = R.rnnt_loss_torch_fast(torch.tensor(acts), torch.tensor(labels),
                                        torch.tensor(al), torch.tensor(ll))
        np.testing.assert_allclose(cf.numpy(), costs, atol=1e-12)
        np.testing.assert_allclose(gf.numpy(), grads, atol=1e-12)


def test_grad_rows_sum_to_zero_inside_and_vanish_outside():
    acts, labels, al, ll = _random_case(5)
    _, grads = R.rnnt_loss(acts, labels, al, ll)
    # softmax chain rule: each cell's gradient over V sums to 0
    assert np.abs(grads.sum(-1)).max() < 1e-12
    for b in range(acts.shape[0]):
        assert np.all(grads[b, al[b]:] == 0)
        assert np.all(grads[b, :, ll[b] + 1:] == 0)
