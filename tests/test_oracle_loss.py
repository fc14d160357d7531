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


# ---------------------------------------------------------------------------------------------
# the DP oracle against the DEFINITION of the loss: explicit enumeration of every monotone alignment
# (oracle/rnnt_loss_bruteforce.py) - nothing of the alpha/beta recursion is shared
def _tiny_case(seed, B, T, U1, V, blank_bias=0.0):
    rng = np.random.default_rng(seed)
    acts = 2.0 * rng.normal(size=(B, T, U1, V))
    acts[..., 0] += blank_bias
    labels = rng.integers(1, V, size=(B, max(U1 - 1, 1))).astype(np.int32)[:, :U1 - 1]
    act_lens = rng.integers(1, T + 1, size=B).astype(np.int32)
    label_lens = rng.integers(0, U1, size=B).astype(np.int32)
    act_lens[0] = T
    label_lens[0] = U1 - 1
    return acts, labels, act_lens, label_lens


def test_alignment_count_is_the_binomial():
    from math import comb
    from oracle import rnnt_loss_bruteforce as BF
    for T in range(1, 6):
        for U in range(0, 5):
            assert sum(1 for _ in BF.alignments(T, U)) == comb(T + U - 1, U)


def test_dp_oracle_equals_sum_over_all_alignments_cost_and_gradient():
    from oracle import rnnt_loss_bruteforce as BF
    n = 0
    for T in (1, 2, 3, 4):
        for U1 in (1, 2, 3, 4):
            for V in (2, 3, 5):
                for seed in range(2):
                    acts, labels, al, ll = _tiny_case(100 * T + 10 * U1 + V + 1000 * seed, 3, T, U1, V,
                                                      blank_bias=(0.0, 2.0)[seed])
                    c_bf, g_bf = BF.rnnt_loss(acts, labels, al, ll)
                    c_dp, g_dp = R.rnnt_loss(acts, labels, al, ll)
                    np.testing.assert_allclose(c_dp, c_bf, rtol=0, atol=1e-11)
                    np.testing.assert_allclose(g_dp, g_bf, rtol=0, atol=1e-11)
                    n += 1
    assert n == 96


def test_bruteforce_reproduces_the_upstream_known_answer():
    from oracle import rnnt_loss_bruteforce as BF
    ka = R.KNOWN_ANSWER
    c, g = BF.rnnt_loss(ka["acts"], ka["labels"], ka["act_lens"], ka["label_lens"])
    assert abs(c[0] - ka["cost"]) < 2e-6
    np.testing.assert_allclose(g, ka["grads"], atol=5e-7)


def test_batch_known_answer_b2_costs_and_gradient():
    """The B = 2 vector of upstream's unit tests (provenance and its limits: oracle/rnnt_loss_ref.py, KNOWN_ANSWER_B2):
    72 logits, 2 published costs and 72 published gradient entries must agree through the float64 recursion AND
    through the enumeration of every alignment - a digit wrong in any of them breaks one of the 148 comparisons."""
    from oracle import rnnt_loss_bruteforce as BF
    ka = R.KNOWN_ANSWER_B2
    costs, grads = R.rnnt_loss(ka["acts"], ka["labels"], ka["act_lens"], ka["label_lens"])
    np.testing.assert_allclose(costs, ka["costs"], rtol=0, atol=5e-7)
    np.testing.assert_allclose(grads, ka["grads"], rtol=0, atol=1e-6)
    c, g = BF.rnnt_loss(ka["acts"], ka["labels"], ka["act_lens"], ka["label_lens"])
    np.testing.assert_allclose(c, ka["costs"], rtol=0, atol=5e-7)
    np.testing.assert_allclose(g, ka["grads"], rtol=0, atol=1e-6)
