"""GPU: the launch-persistent forward (csrc/stack_kernels.hip stack_fwd_lpw_kernel + forward_lpw in
csrc/encoder_stack.hip; EDGEDICT_STACK_LPW=1) and the split-K weights-stationary BPTT (stack_bwd_sk_kernel) of the
encoder stack against the launch-per-step kernels.

One launch carries several CONSECUTIVE time steps of every runnable layer; workgroups keep W_hh in registers,
meet through arrival counters and exchange h with write-through stores / L2-served loads inside the launch.
The arithmetic is the step kernel's, so EVERYTHING must be bit-identical: outputs, final states, the saved
gates / cell states (checked through the weight gradients the unchanged backward pass computes from them).
A stale read of another workgroup's h, a missed counter or a wrong image parity shows up as a mismatch.
Reference arithmetic: ResLayerNormLSTM.forward rnnt/models.py:55-75."""
import os

import pytest
import torch

from test_encoder_stack_gpu import CASES, _encoder, _run

pytestmark = pytest.mark.gpu


def _with_env(fn, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _same(a, b, exact_bias=False):
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for n in a[3]:
        # column sums use fp32 atomics; the output projection's weight gradient is a split-K product with atomic
        # accumulation when the batch is large enough (B = 64 x H = 512 here) - neither goes through the recurrence
        if "norm" in n or "proj" in n or "bias" in n:
            scale = max(a[3][n].abs().max().item(), 1e-6)
            assert (a[3][n] - b[3][n]).abs().max().item() <= 1e-4 * scale, n
        else:
            assert torch.equal(a[3][n], b[3][n]), n


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("steps", [2, 4])
@pytest.mark.parametrize("poll", [1, 0])
def test_lpw_forward_is_bit_identical_to_the_step_kernels(hip_lib, case, steps, poll):
    """poll = 1 (default): no counter on the dependency chain - the readers gather and recognise chunks that are not
    written yet by the fill pattern (EDGEDICT_LPW_POLL); 0: the readers poll the arrival counters.  EDGEDICT_STACK_POISON
    fills the per-frame images with NaN also in the counter mode, so a read that overtakes its producer cannot pass on
    the previous run's values."""
    from edgedict_amd import encoder_stack
    chunk = 4 if case[6] < 4 else case[6]        # the steps per launch divide the chunk
    enc, xs = _encoder(case)
    ref = _with_env(lambda: _run(enc, xs, torch.bfloat16, flags=0, chunk=chunk, lag=case[7]), EDGEDICT_STACK_LPW=0)
    got = _with_env(lambda: _run(enc, xs, torch.bfloat16, flags=0, chunk=chunk, lag=case[7]),
                    EDGEDICT_STACK_LPW=1, EDGEDICT_LPW_STEPS=steps, EDGEDICT_LPW_POLL=poll, EDGEDICT_STACK_POISON=1)
    ser = _with_env(lambda: _run(enc, xs, torch.bfloat16, flags=encoder_stack.SERIAL, chunk=chunk, lag=case[7]),
                    EDGEDICT_STACK_LPW=1, EDGEDICT_LPW_STEPS=steps, EDGEDICT_LPW_POLL=poll, EDGEDICT_STACK_POISON=1)
    _same(ref, got)
    _same(ref, ser)
    encoder_stack.check_wsr_error()


# larger batches with ragged last tiles, H % 256 == 0, and B > 64 (two row groups per layer in the forward kernel)
BIG_CASES = [(64, 21, 16, 128, 3, [1], 4, 0), (37, 19, 24, 64, 2, [0], 3, 0), (49, 16, 16, 256, 2, [], 4, 0),
             (100, 11, 16, 64, 2, [1], 2, 0)]


@pytest.mark.parametrize("case", BIG_CASES)
@pytest.mark.parametrize("poll", [1, 0])
def test_lpw_forward_bigger_batches_bit_identical(hip_lib, case, poll):
    from edgedict_amd import encoder_stack
    chunk = 4 if case[6] < 4 else case[6]
    enc, xs = _encoder(case)
    ref = _with_env(lambda: _run(enc, xs, torch.bfloat16, flags=0, chunk=chunk), EDGEDICT_STACK_LPW=0)
    got = _with_env(lambda: _run(enc, xs, torch.bfloat16, flags=0, chunk=chunk),
                    EDGEDICT_STACK_LPW=1, EDGEDICT_LPW_STEPS=chunk, EDGEDICT_LPW_POLL=poll, EDGEDICT_STACK_POISON=1)
    _same(ref, got)
    encoder_stack.check_wsr_error()


@pytest.mark.parametrize("poll", [1, 0])
def test_lpw_forward_e6d2_full_size_bit_identical_and_carried_state(hip_lib, poll, monkeypatch):
    """BASELINE config 2's encoder (B = 64, T0 = 401, 6 x 1024, 2x time reduction): 4 layer slots of 64
    workgroups fill the chip, the DEFAULT chunk / steps per launch; then chunked evaluation with carried state."""
    from edgedict_amd import config, encoder_stack
    monkeypatch.setenv("EDGEDICT_LPW_POLL", str(poll))
    chunk = encoder_stack.CHUNK
    case = (64, 401, 240, 1024, 6, [1], chunk, 0)
    enc, xs = _encoder(case)
    ref = _with_env(lambda: _run(enc, xs, torch.bfloat16, flags=0, chunk=chunk), EDGEDICT_STACK_LPW=0)
    got = _with_env(lambda: _run(enc, xs, torch.bfloat16, flags=0, chunk=chunk), EDGEDICT_STACK_LPW=1, EDGEDICT_STACK_POISON=1)
    assert encoder_stack.last_mode(False) == (1, chunk)
    assert got[0].shape == (64, 201, 24)
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1]) and torch.equal(ref[2], got[2])
    for n in ref[3]:
        if "weight_ih" in n or "weight_hh" in n:
            assert torch.isfinite(got[3][n]).all(), n
    encoder_stack.check_wsr_error()

    def chunked():
        enc.compute_dtype = torch.bfloat16
        with torch.no_grad():
            full, (hf, cf) = enc(xs[:8])
            y1, (h1, c1) = enc(xs[:8, :200])
            y2, (h2, c2) = enc(xs[:8, 200:], (h1, c1))
        return full, hf, cf, torch.cat([y1, y2], 1), h2, c2
    full, hf, cf, cat, h2, c2 = _with_env(chunked, EDGEDICT_STACK_LPW=1)
    assert torch.equal(cat, full) and torch.equal(h2, hf) and torch.equal(c2, cf)


def _close(a, b, tol):
    """outputs / states identical (the forward pass is untouched); every gradient within `tol` of its norm."""
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    worst = 0.0
    for n in a[3]:
        d = (a[3][n].double() - b[3][n].double()).norm().item() / max(a[3][n].double().norm().item(), 1e-12)
        worst = max(worst, d)
        assert d < tol, (n, d)
    return worst


SK_CASES = [c for c in CASES if c[3] % 64 == 0 and c[0] <= 64] + [
    (7, 21, 32, 128, 3, [1], 4, 0), (33, 30, 16, 64, 2, [0], 6, 0),
    (9, 13, 32, 256, 2, [0], 4, 0),       # H % 256 == 0: per-quarter arrival counters, 4 unit blocks
    (64, 10, 16, 512, 2, [], 4, 0),       # 8 unit blocks, full batch
    (5, 11, 16, 192, 2, [1], 4, 0),       # 6 k-steps per quarter: a partial ring slot, rotated walk over 2 groups
    (40, 9, 16, 320, 1, [], 4, 0),        # 10 k-steps per quarter, 3 row tiles
]


@pytest.mark.parametrize("case", SK_CASES + [c for c in BIG_CASES if c[0] <= 64])
@pytest.mark.parametrize("steps", [2, 4])
def test_split_k_bptt_matches_the_step_kernels(hip_lib, case, steps):
    """stack_bwd_sk_kernel (EDGEDICT_STACK_BWD_SK=1): a workgroup owns 64 units x one quarter of the 4H gate columns,
    W_hh^T stationary in registers, partial sums exchanged between the 4 workgroups of a unit block.  The K split
    changes the order of the fp32 sums, so dG differs from the step kernels' in bf16 rounding only: every parameter
    gradient within 1e-2 of its norm (measured ~2e-3), deterministic (two runs bit-identical), serial == multi-stream.
    EDGEDICT_STACK_POISON: the dG images and the partial buffers hold NaN when the pass starts, so a read that overtakes
    its producer cannot pass on the previous run's values."""
    from edgedict_amd import encoder_stack
    chunk = 4 if case[6] < 4 else case[6]
    enc, xs = _encoder(case)
    ref = _with_env(lambda: _run(enc, xs, torch.bfloat16, flags=0, chunk=chunk, lag=case[7]), EDGEDICT_STACK_BWD_SK=0)
    got = _with_env(lambda: _run(enc, xs, torch.bfloat16, flags=0, chunk=chunk, lag=case[7]),
                    EDGEDICT_STACK_BWD_SK=1, EDGEDICT_SK_STEPS=steps, EDGEDICT_STACK_POISON=1)
    again = _with_env(lambda: _run(enc, xs, torch.bfloat16, flags=0, chunk=chunk, lag=case[7]),
                      EDGEDICT_STACK_BWD_SK=1, EDGEDICT_SK_STEPS=steps, EDGEDICT_STACK_POISON=1)
    ser = _with_env(lambda: _run(enc, xs, torch.bfloat16, flags=encoder_stack.SERIAL, chunk=chunk, lag=case[7]),
                    EDGEDICT_STACK_BWD_SK=1, EDGEDICT_SK_STEPS=steps, EDGEDICT_STACK_POISON=1)
    worst = _close(ref, got, 1e-2)
    _same(got, again)
    _same(got, ser)
    encoder_stack.check_wsr_error()
    print("\n[split-K BPTT %s steps %d] worst gradient deviation from the step kernels: %.2e of the norm" % (case, steps, worst))


def test_split_k_bptt_e6d2_full_size(hip_lib):
    """The benched geometry (B = 64, T0 = 401, 6 x 1024, 2x time reduction; 16 unit blocks x 4 quarters = 64
    workgroups per layer, 4 layers per launch, per-quarter counters) at the DEFAULT chunk / steps per launch: every
    parameter gradient within 1e-2 of its norm of the launch-per-step kernels' (the K split re-orders fp32 sums, dG is
    re-rounded to bf16 on each of the 401 steps; measured: 3.3e-3 for the input LayerNorm's gain, the most sensitive
    one), run-to-run bit-identical (the second run from NaN-filled images and partials), no bounded wait gave up."""
    from edgedict_amd import encoder_stack
    chunk = encoder_stack.CHUNK
    case = (64, 401, 240, 1024, 6, [1], chunk, 0)
    enc, xs = _encoder(case)
    ref = _with_env(lambda: _run(enc, xs, torch.bfloat16, flags=0, chunk=chunk), EDGEDICT_STACK_BWD_SK=0)
    got = _with_env(lambda: _run(enc, xs, torch.bfloat16, flags=0, chunk=chunk), EDGEDICT_STACK_BWD_SK=1)
    assert encoder_stack.last_mode(True) == (2, chunk)
    again = _with_env(lambda: _run(enc, xs, torch.bfloat16, flags=0, chunk=chunk), EDGEDICT_STACK_BWD_SK=1, EDGEDICT_STACK_POISON=1)
    worst = _close(ref, got, 1e-2)
    _same(got, again)
    encoder_stack.check_wsr_error()
    print("\n[split-K BPTT, E6D2 size] worst gradient deviation from the step kernels: %.2e of the norm" % worst)
