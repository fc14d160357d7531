"""GPU parity: HIP RNN-T loss (through the C ABI) vs the float64 oracle."""
import numpy as np
import pytest
import torch

from oracle import rnnt_loss_ref as R

pytestmark = pytest.mark.gpu


def _case(seed, B, T, U1, V, ragged=True, scale=1.0):
    rng = np.random.default_rng(seed)
    acts = (scale * rng.normal(size=(B, T, U1, V))).astype(np.float32)
    labels = rng.integers(1, V, size=(B, max(U1 - 1, 0))).astype(np.int32)
    if ragged:
        al = rng.integers(1, T + 1, size=B).astype(np.int32)
        ll = rng.integers(0, U1, size=B).astype(np.int32)
    else:
        al = np.full(B, T, np.int32)
        ll = np.full(B, U1 - 1, np.int32)
    al[0] = T
    ll[0] = U1 - 1
    return acts, labels, al, ll


def _run_hip(acts, labels, al, ll, dtype=torch.float32, reduction="mean", blank=0):
    from edgedict_amd.loss import RNNTLoss
    dev = "cuda:0"
    ta = torch.tensor(acts, device=dev).to(dtype).requires_grad_(True)
    loss = RNNTLoss(blank=blank, reduction=reduction)(
        ta, torch.tensor(labels, device=dev), torch.tensor(al, device=dev),
        torch.tensor(ll, device=dev))
    (loss.sum() if reduction == "none" else loss).backward()
    return loss.detach().float().cpu().numpy(), ta.grad.float().cpu().numpy()


def test_known_answer(hip_lib):
    ka = R.KNOWN_ANSWER
    loss, grads = _run_hip(ka["acts"].astype(np.float32), ka["labels"], ka["act_lens"],
                           ka["label_lens"], reduction="sum")
    assert loss.shape == (1,)
    assert abs(loss[0] - ka["cost"]) < 1e-5
    np.testing.assert_allclose(grads, ka["grads"], atol=2e-6)


def test_batch_known_answer_b2(hip_lib):
    """Upstream's B = 2 unit-test vector (oracle/rnnt_loss_ref.py, KNOWN_ANSWER_B2): both costs and all 72 gradient
    entries, reduction 'none' / 'sum' (gradient of the sum) and 'mean' (x 1/B)."""
    ka = R.KNOWN_ANSWER_B2
    loss, grads = _run_hip(ka["acts"].astype(np.float32), ka["labels"], ka["act_lens"], ka["label_lens"],
                           reduction="none")
    np.testing.assert_allclose(loss, ka["costs"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(grads, ka["grads"], rtol=0, atol=2e-6)
    loss_m, grads_m = _run_hip(ka["acts"].astype(np.float32), ka["labels"], ka["act_lens"], ka["label_lens"])
    assert abs(loss_m[0] - ka["costs"].mean()) < 2e-6
    np.testing.assert_allclose(grads_m, ka["grads"] / 2, rtol=0, atol=2e-6)


@pytest.mark.parametrize("B,T,U1,V,ragged", [
    (1, 1, 1, 7, False),       # single cell, empty label sequence
    (2, 5, 1, 16, True),       # U = 0 for every utterance
    (3, 7, 5, 11, True),       # V not a multiple of the vector width (scalar path)
    (4, 33, 9, 64, True),
    (2, 40, 70, 128, True),    # U+1 > 64: two wavefronts per lattice
    (4, 84, 21, 2048, True),   # E4D1 / 5 s lattice
])
def test_fp32_matches_oracle(hip_lib, B, T, U1, V, ragged):
    acts, labels, al, ll = _case(B * 1000 + T, B, T, U1, V, ragged)
    costs, grads = R.rnnt_loss(acts.astype(np.float64), labels, al, ll)
    loss, g = _run_hip(acts, labels, al, ll, reduction="none")
    np.testing.assert_allclose(loss, costs, rtol=1e-5, atol=1e-4)
    # alpha/beta live in fp32 log space (as in upstream's GPU path): exp(a+b-ll) cancels
    # ~|ll| of magnitude, so allow rtol ~ |ll| * 2^-23 on top of the absolute floor.
    np.testing.assert_allclose(g, grads, rtol=1e-3, atol=2e-5)
    # mean reduction = sum/B with shape (1,), gradient scaled by 1/B (rnnt/models.py:238 use)
    loss_m, g_m = _run_hip(acts, labels, al, ll, reduction="mean")
    assert loss_m.shape == (1,)
    np.testing.assert_allclose(loss_m[0], costs.mean(), rtol=1e-5)
    np.testing.assert_allclose(g_m, grads / B, rtol=1e-3, atol=2e-5)


def test_large_logit_range_is_stable(hip_lib):
    acts, labels, al, ll = _case(7, 2, 12, 6, 32, scale=30.0)
    costs, grads = R.rnnt_loss(acts.astype(np.float64), labels, al, ll)
    loss, g = _run_hip(acts, labels, al, ll, reduction="none")
    assert np.isfinite(loss).all() and np.isfinite(g).all()
    np.testing.assert_allclose(loss, costs, rtol=2e-5)
    np.testing.assert_allclose(g, grads, atol=5e-4)  # |ll| ~ 1e3 in fp32 log space


def test_bf16_logits(hip_lib):
    acts, labels, al, ll = _case(3, 3, 20, 8, 256)
    acts_bf = torch.tensor(acts).bfloat16().float().numpy()  # oracle sees the rounded logits
    costs, grads = R.rnnt_loss(acts_bf.astype(np.float64), labels, al, ll)
    loss, g = _run_hip(acts, labels, al, ll, dtype=torch.bfloat16, reduction="none")
    np.testing.assert_allclose(loss, costs, rtol=1e-4)
    np.testing.assert_allclose(g, grads, atol=4e-3)  # bf16 output rounding (8 mantissa bits)


def test_debug_views_alpha_beta(hip_lib):
    from edgedict_amd.loss import rnnt_loss_debug
    acts, labels, al, ll = _case(11, 2, 9, 4, 16)
    dev = "cuda:0"
    costs, denom, alphas, betas, lls = rnnt_loss_debug(
        torch.tensor(acts, device=dev), torch.tensor(labels, device=dev),
        torch.tensor(al, device=dev), torch.tensor(ll, device=dev))
    for b in range(2):
        T, U = int(al[b]), int(ll[b])
        lp = R.log_softmax(acts[b, :T, :U + 1].astype(np.float64))
        alpha, beta, loglike = R.lattice(lp, labels[b], T, U)
        np.testing.assert_allclose(alphas[b, :T, :U + 1].cpu().numpy(), alpha, atol=1e-4)
        np.testing.assert_allclose(betas[b, :T, :U + 1].cpu().numpy(), beta, atol=1e-4)
        # alpha-side and beta-side log-likelihoods agree (size-independent property)
        assert abs(lls[b, 0].item() - lls[b, 1].item()) < 1e-3
        assert abs(lls[b, 0].item() - loglike) < 1e-3


def test_full_size_lattice_properties(hip_lib):
    """E6D2 / B=8 slice of the BASELINE lattice (T'=201, U+1=65, V=2048): too big for the
    pure-Python oracle, so check size-independent properties instead."""
    from edgedict_amd.loss import rnnt_loss_debug, RNNTLoss
    B, T, U1, V = 8, 201, 65, 2048
    g = torch.Generator(device="cpu").manual_seed(0)
    acts = torch.randn(B, T, U1, V, generator=g).cuda()
    labels = torch.randint(4, V, (B, U1 - 1), generator=g, dtype=torch.int32).cuda()
    al = torch.randint(150, T + 1, (B,), generator=g, dtype=torch.int32)
    ll = torch.randint(32, U1, (B,), generator=g, dtype=torch.int32)
    al[0], ll[0] = T, U1 - 1
    al, ll = al.cuda(), ll.cuda()
    costs, denom, alphas, betas, lls = rnnt_loss_debug(acts, labels, al, ll)
    assert torch.isfinite(costs).all()
    assert (lls[:, 0] - lls[:, 1]).abs().max().item() < 2e-2 * 1  # alpha(T-1,U)+lpb == beta(0,0)
    rel = ((lls[:, 0] - lls[:, 1]).abs() / lls[:, 0].abs()).max().item()
    assert rel < 1e-5
    a = acts.clone().requires_grad_(True)
    loss = RNNTLoss(reduction="sum")(a, labels, al, ll)
    loss.backward()
    gr = a.grad
    assert gr.sum(-1).abs().max().item() < 1e-4          # rows sum to zero
    for b in range(B):
        assert gr[b, int(al[b]):].abs().max().item() == 0 if int(al[b]) < T else True
        assert gr[b, :, int(ll[b]) + 1:].abs().max().item() == 0 if int(ll[b]) + 1 < U1 else True
    # cross-check against the vectorised torch port on one utterance (fp64 on CPU)
    cf, gf = R.rnnt_loss_torch_fast(acts[:1].double().cpu(), labels[:1].cpu(), al[:1].cpu(),
                                    ll[:1].cpu())
    assert abs(cf[0].item() - costs[0].item()) / cf[0].item() < 1e-5
    assert (gf[0].float() - gr[0].cpu()).abs().max().item() < 2e-5


def test_fused_lse_epilogue_matches_separate_pass(hip_lib):
    """The logits product with log-softmax partials in its epilogue (edgedict_gemm_nt_lse +
    edgedict_rnnt_loss_forward_packed_parts) against the plain product followed by the pass over the
    logits (edgedict_rnnt_loss_forward_packed): identical logits, denominators equal up to the summation
    order, costs within 1e-6 relative - and the model-level loss / gradients do not move."""
    import ctypes
    from edgedict_amd import _lib, config
    from edgedict_amd.ops import _ll
    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(3)
    B, T, U1, V, J = 5, 37, 9, 2048, 640
    act = torch.tensor([37, 30, 12, 37, 5], dtype=torch.int32)
    lab = torch.tensor([8, 3, 8, 0, 6], dtype=torch.int32)
    rows = act.long() * (lab.long() + 1)
    off = torch.zeros(B, dtype=torch.int64)
    off[1:] = torch.cumsum(rows, 0)[:-1]
    M = int(rows.sum())
    hid = torch.tanh(torch.randn(M, J, generator=g)).to(torch.bfloat16).cuda()
    w2 = (torch.randn(V, J, generator=g) / 8).to(torch.bfloat16).cuda()
    b2 = torch.randn(V, generator=g).cuda()
    labels = torch.randint(1, V, (B, U1 - 1), generator=g, dtype=torch.int32).cuda()
    act_d, lab_d, off_d = act.cuda(), lab.cuda(), off.cuda()
    ws_bytes = lib.edgedict_rnnt_workspace_bytes(B, T, U1)

    def run(fused):
        ws = torch.zeros(ws_bytes, dtype=torch.uint8, device="cuda")
        costs = torch.empty(B, device="cuda")
        red = torch.empty(1, device="cuda")
        logits = torch.empty(M, V, dtype=torch.bfloat16, device="cuda")
        if fused:
            parts = torch.empty(M, V // 64, 2, device="cuda")
            _lib.call("gemm_nt_lse", hid, _ll(J), w2, _ll(J), logits, _ll(V), M, V, J, b2, parts)
            _lib.call("rnnt_loss_forward_packed_parts", logits, labels, act_d, lab_d, off_d, B, T, U1, V, 0,
                      costs, red, 1.0 / B, ws, parts, V // 64)
        else:
            _lib.call("gemm", 1, 1, hid, _ll(J), 1, w2, _ll(J), 1, logits, _ll(V), M, V, J, b2, None, 0, 1)
            _lib.call("rnnt_loss_forward_packed", logits, 1, labels, act_d, lab_d, off_d, B, T, U1, V, 0,
                      costs, red, 1.0 / B, ws)
        torch.cuda.synchronize()
        n = B * T * U1
        denom = torch.frombuffer(ws.cpu().numpy().tobytes()[:4 * n], dtype=torch.float32).clone()
        return logits.float().cpu(), denom.view(B, T, U1), costs.cpu(), red.item()

    lf, df, cf, rf = run(True)
    lp, dp, cp, rp = run(False)
    # the 128x128 / vendor kernels and the 256x256 kernel sum in different orders: one bf16 ulp at most
    assert ((lf - lp).abs() <= 2.0 ** -7 * lp.abs() + 1e-2).all()
    ref = lf.double().logsumexp(dim=1)          # denominators of the FUSED run's own stored logits
    k = 0
    for b in range(B):
        for t in range(int(act[b])):
            for u in range(int(lab[b]) + 1):
                assert abs(df[b, t, u].item() - ref[k].item()) < 2e-5, (b, t, u)
                k += 1
    assert ((cf - cp).abs() <= 2e-3 * cp.abs()).all()       # different logits rounding (summation order)
    # same logits -> same costs to 1e-6: run the separate pass on the fused run's logits
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device="cuda")
    costs = torch.empty(B, device="cuda")
    red = torch.empty(1, device="cuda")
    _lib.call("rnnt_loss_forward_packed", lf.to(torch.bfloat16).cuda(), 1, labels, act_d, lab_d, off_d, B, T, U1,
              V, 0, costs, red, 1.0 / B, ws)
    assert ((costs.cpu() - cf).abs() <= 1e-6 * cf.abs()).all(), (costs.cpu(), cf)


def test_hip_loss_equals_the_sum_over_all_alignments(hip_lib):
    """The HIP kernels against the DEFINITION of the loss (oracle/rnnt_loss_bruteforce.py: every monotone
    alignment enumerated, -log of the summed path probabilities, autograd gradient) on tiny ragged
    lattices - no alpha/beta recursion on the checking side."""
    from oracle import rnnt_loss_bruteforce as BF
    for T, U1, V, seed in [(1, 1, 2, 0), (2, 3, 5, 1), (3, 2, 3, 2), (4, 4, 5, 3), (4, 3, 2, 4), (3, 4, 4, 5)]:
        rng = np.random.default_rng(900 + seed)
        B = 4
        acts = (2.0 * rng.normal(size=(B, T, U1, V))).astype(np.float32)
        labels = rng.integers(1, V, size=(B, max(U1 - 1, 1))).astype(np.int32)[:, :U1 - 1]
        al = rng.integers(1, T + 1, size=B).astype(np.int32)
        ll = rng.integers(0, U1, size=B).astype(np.int32)
        al[0], ll[0] = T, U1 - 1
        c_bf, g_bf = BF.rnnt_loss(acts.astype(np.float64), labels, al, ll)
        loss, g = _run_hip(acts, np.ascontiguousarray(labels), al, ll, reduction="none")
        np.testing.assert_allclose(loss, c_bf, rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(g, g_bf, rtol=0, atol=5e-6)


@pytest.mark.gpu
def test_backward_in_utterance_ranges_equals_one_pass(hip_lib):
    """edgedict_rnnt_loss_backward_packed_range over [0,2), [2,2), [2,5) writes exactly the gradient of the one-pass
    entry point (same kernel, utterance index offset by b0): bit-identical, rows of other utterances untouched."""
    from edgedict_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(11)
    B, T, U1, V = 5, 23, 7, 256
    act = torch.tensor([23, 20, 9, 23, 4], dtype=torch.int32)
    lab = torch.tensor([6, 2, 6, 0, 5], dtype=torch.int32)
    rows = act.long() * (lab.long() + 1)
    off = torch.zeros(B, dtype=torch.int64)
    off[1:] = torch.cumsum(rows, 0)[:-1]
    M = int(rows.sum())
    for dt, code in ((torch.float32, 0), (torch.bfloat16, 1)):
        logits = torch.randn(M, V, generator=g).to(dt).cuda()
        labels = torch.randint(1, V, (B, U1 - 1), generator=g, dtype=torch.int32).cuda()
        act_d, lab_d, off_d = act.cuda(), lab.cuda(), off.cuda()
        ws = torch.zeros(lib.edgedict_rnnt_workspace_bytes(B, T, U1), dtype=torch.uint8, device="cuda")
        costs, red = torch.empty(B, device="cuda"), torch.empty(1, device="cuda")
        _lib.call("rnnt_loss_forward_packed", logits, code, labels, act_d, lab_d, off_d, B, T, U1, V, 0, costs, red,
                  1.0 / B, ws)
        one = torch.zeros_like(logits)
        _lib.call("rnnt_loss_backward_packed", logits, code, one, labels, act_d, lab_d, off_d, B, T, U1, V, 0, ws,
                  1.0 / B, None, 0)
        parts = torch.full_like(logits, 7.0)
        for b0, nb in ((0, 2), (2, 0), (2, 3)):
            _lib.call("rnnt_loss_backward_packed_range", logits, code, parts, labels, act_d, lab_d, off_d, B, T, U1, V,
                      0, ws, 1.0 / B, None, 0, b0, nb)
            if (b0, nb) == (0, 2):
                torch.cuda.synchronize()
                assert (parts[int(off[2]):] == 7.0).all()        # utterances 2.. not touched yet
        torch.cuda.synchronize()
        assert torch.equal(one, parts)
    with pytest.raises(RuntimeError):
        _lib.call("rnnt_loss_backward_packed_range", logits, code, parts, labels, act_d, lab_d, off_d, B, T, U1, V, 0, ws,
                  1.0 / B, None, 0, 3, 3)


@pytest.mark.gpu
def test_fused_column_sums_of_the_gradient(hip_lib):
    """edgedict_rnnt_loss_backward_packed_colsum: the gradient matrix is the plain entry point's bit for bit, and the partial
    rows it leaves add up to the column sums of that matrix (of the fp32 values in front of the store: in bf16 they differ
    from the sums of the stored values by the rounding, 2^-9 per element) - the joint's output-bias gradient without a
    second pass over the matrix.  V at the limit (2048 bf16 / 1024 f32), ragged boxes, an empty one."""
    from edgedict_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(13)
    B, T, U1 = 5, 23, 7
    act = torch.tensor([23, 20, 9, 23, 4], dtype=torch.int32)
    lab = torch.tensor([6, 2, 6, 0, 5], dtype=torch.int32)
    rows = act.long() * (lab.long() + 1)
    off = torch.zeros(B, dtype=torch.int64)
    off[1:] = torch.cumsum(rows, 0)[:-1]
    M = int(rows.sum())
    for dt, code, V in ((torch.float32, 0, 1024), (torch.bfloat16, 1, 2048), (torch.bfloat16, 1, 264)):
        logits = torch.randn(M, V, generator=g).to(dt).cuda()
        labels = torch.randint(1, V, (B, U1 - 1), generator=g, dtype=torch.int32).cuda()
        act_d, lab_d, off_d = act.cuda(), lab.cuda(), off.cuda()
        ws = torch.zeros(lib.edgedict_rnnt_workspace_bytes(B, T, U1), dtype=torch.uint8, device="cuda")
        costs, red = torch.empty(B, device="cuda"), torch.empty(1, device="cuda")
        _lib.call("rnnt_loss_forward_packed", logits, code, labels, act_d, lab_d, off_d, B, T, U1, V, 0, costs, red,
                  1.0 / B, ws)
        plain = torch.zeros_like(logits)
        _lib.call("rnnt_loss_backward_packed", logits, code, plain, labels, act_d, lab_d, off_d, B, T, U1, V, 0, ws,
                  1.0 / B, None, 0)
        n = lib.edgedict_rnnt_grad_colsum_rows(code, B, T, U1, V)
        assert n > 0
        parts = torch.full((n, V), float("nan"), device="cuda")
        fused = torch.zeros_like(logits)
        _lib.call("rnnt_loss_backward_packed_colsum", logits, code, fused, labels, act_d, lab_d, off_d, B, T, U1, V, 0,
                  ws, 1.0 / B, None, 0, parts)
        torch.cuda.synchronize()
        assert torch.equal(plain, fused)
        want = plain.double().sum(0)
        got = parts.double().sum(0)
        assert torch.isfinite(got).all()
        tol = 1e-5 if dt == torch.float32 else 2.0 ** -8
        assert ((got - want).abs() <= tol * plain.double().abs().sum(0) + 1e-12).all()
    assert lib.edgedict_rnnt_grad_colsum_rows(1, B, T, U1, 2056) == 0          # more than four vector passes of a wave
    assert lib.edgedict_rnnt_grad_colsum_rows(1, B, T, U1, 100) == 0           # rows not 16-byte multiples
    with pytest.raises(RuntimeError):
        _lib.call("rnnt_loss_backward_packed_colsum", logits, code, fused, labels, act_d, lab_d, off_d, B, T, U1, V, 0, ws,
                  1.0 / B, None, 0, None)


@pytest.mark.gpu
def test_long_lattice_drift_is_bounded(hip_lib):
    """a chain of T + U ~ 1200 log-adds (T = 1000, U = 200; peaked logits, so most log-adds see a tiny
    second term - the regime in which a correction term evaluated as log(1 + x) drops x altogether).  Cost and the
    whole gradient against the float64 restatement: the cost within 2e-6 relative (north-star bound: 1e-3), the
    gradient within 2e-5 absolute (the bound of the short-lattice cases)."""
    rng = np.random.default_rng(123)
    B, T, U1, V = 1, 1000, 201, 16
    acts = (4.0 * rng.normal(size=(B, T, U1, V))).astype(np.float32)
    labels = rng.integers(1, V, size=(B, U1 - 1)).astype(np.int32)
    al = np.array([T], np.int32)
    ll = np.array([U1 - 1], np.int32)
    costs, grads = R.rnnt_loss(acts.astype(np.float64), labels, al, ll)
    loss, g = _run_hip(acts, labels, al, ll, reduction="none")
    assert abs(loss[0] - costs[0]) <= 2e-6 * abs(costs[0]), (loss[0], costs[0])
    assert np.abs(g - grads).max() <= 2e-5, np.abs(g - grads).max()
