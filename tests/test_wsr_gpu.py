"""GPU: the weights-stationary recurrence path of the encoder stack (csrc/wsr_kernels.hip; H = 1024,
B <= 64: one launch per CHUNK of frames, W_hh in registers, h_t exchanged through the XCD's L2)
against the launch-per-step kernels it replaces (csrc/stack_kernels.hip) - the same arithmetic
(rnnt/models.py:55-75) in a different summation order - and against itself on one stream.
The reference-pinned checks of this path are tests/test_e6d2_parity_gpu.py (bench configuration)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# (B, T0, L, time_reductions, chunk)
CASES = [
    (2, 26, 2, [1], 3),
    (5, 31, 3, [1], 2),
    (33, 24, 2, [0], 4),
    (64, 50, 3, [1], 6),
]


def _encoder(B, T0, L, red, seed=0):
    from edgedict_amd.models import Encoder
    torch.manual_seed(seed)
    enc = Encoder(input_size=240, hidden_size=1024, num_layers=L, dropout=0.0, proj_size=64,
                  time_reductions=red)
    with torch.no_grad():
        for p in enc.parameters():
            if p.dim() == 1 and p.numel() in (240, 1024):
                p.add_(0.2 * torch.randn_like(p))
    enc = enc.cuda()
    enc.compute_dtype = torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    return enc, torch.randn(B, T0, 240, generator=g).cuda()


def _run(enc, xs, flags, chunk):
    from edgedict_amd import config, encoder_stack
    old = (encoder_stack.CHUNK, encoder_stack.FLAGS, encoder_stack.SPLIT_K, config.STACK_MIN_FRAMES)
    encoder_stack.CHUNK, encoder_stack.FLAGS, encoder_stack.SPLIT_K = chunk, flags, 1
    config.STACK_MIN_FRAMES = 1
    try:
        enc.zero_grad(set_to_none=True)
        out, (h, c) = enc(xs)
        g = torch.Generator(device="cpu").manual_seed(9)
        w = torch.randn(out.shape, generator=g).cuda()
        (out.float() * w).sum().backward()
        torch.cuda.synchronize()
        encoder_stack.check_wsr_error()
        return out.detach().float(), h.detach(), c.detach(), {n: p.grad.clone() for n, p in enc.named_parameters()}
    finally:
        encoder_stack.CHUNK, encoder_stack.FLAGS, encoder_stack.SPLIT_K, config.STACK_MIN_FRAMES = old


def _nrel(a, b):
    return (a.double() - b.double()).norm().item() / max(b.double().norm().item(), 1e-30)


@pytest.mark.parametrize("persist", ["1", "0"])
@pytest.mark.parametrize("case", CASES)
def test_wsr_forward_matches_step_kernels_and_feeds_their_backward(hip_lib, case, persist, monkeypatch):
    """persist=1: ONE launch, layers free-running on their XCDs, LayerNorm + input products by the worker
    workgroups inside the launch; persist=0: one launch per chunk, side work on the side stream."""
    from edgedict_amd import encoder_stack
    monkeypatch.setenv("EDGEDICT_WSR_PERSIST", persist)
    B, T0, L, red, chunk = case
    enc, xs = _encoder(B, T0, L, red)
    new = _run(enc, xs, encoder_stack.WSR, chunk)
    old = _run(enc, xs, 0, chunk)
    assert new[0].shape == old[0].shape and torch.isfinite(new[0]).all()
    # same products, different fp32 summation order, bf16 activations: small, not zero
    assert _nrel(new[0], old[0]) < 5e-3, _nrel(new[0], old[0])
    assert _nrel(new[1], old[1]) < 5e-3 and _nrel(new[2], old[2]) < 5e-3
    # the BPTT kernels read the gates the forward pass left in G: gradients agree as well
    for n in old[3]:
        assert _nrel(new[3][n], old[3][n]) < 4e-2, (n, _nrel(new[3][n], old[3][n]))


@pytest.mark.parametrize("case", CASES[:2] + CASES[3:])
def test_wsr_multi_stream_schedule_is_bit_exact_vs_one_stream(hip_lib, case):
    from edgedict_amd import encoder_stack
    B, T0, L, red, chunk = case
    enc, xs = _encoder(B, T0, L, red)
    a = _run(enc, xs, encoder_stack.WSR, chunk)
    b = _run(enc, xs, encoder_stack.WSR | encoder_stack.SERIAL, chunk)
    c = _run(enc, xs, encoder_stack.WSR, chunk + 1)          # chunking does not change any value
    for other in (b, c):
        assert torch.equal(a[0], other[0]) and torch.equal(a[1], other[1]) and torch.equal(a[2], other[2])


def test_wsr_dry_run_schedule_covers_every_frame_once(hip_lib):
    """edgedict_stack_schedule with the WSR flag: every (layer, frame) is carried by exactly one launch,
    in order, a layer's chunk never before the launch that finished its input chunk, <= L slots."""
    import os
    import numpy as np
    from edgedict_amd import encoder_stack
    old = encoder_stack.FLAGS
    os.environ["EDGEDICT_WSR_DELAY"] = "1"
    os.environ["EDGEDICT_WSR_PERSIST"] = "0"
    try:
        for T0, chunk in ((401, 12), (401, 6), (251, 5), (37, 1)):
            steps, enq, n, slots = encoder_stack.schedule(T0, 240, 1024, [1, 2, 1, 1, 1, 1], B=64, chunk=chunk,
                                                          flags=encoder_stack.WSR)
            assert slots <= 6
            for l, s in enumerate(steps):
                assert (s >= 0).all() and (np.diff(s) >= 0).all(), l
            f = [2, 2, 1, 1, 1, 1]
            for l in range(1, 6):
                cf_prev, cf = chunk * f[l - 1], chunk * f[l]
                for k in range(len(enq[l])):
                    # chunk k of layer l starts after the launch that finished chunk k of layer l-1
                    last_prev = steps[l - 1][min(len(steps[l - 1]), (k + 1) * cf_prev) - 1]
                    assert steps[l][k * cf] > last_prev, (l, k)
            nch0 = (T0 + 2 * chunk - 1) // (2 * chunk)
            assert n == nch0 + 5, (T0, chunk, n)     # wavefront: one extra launch per layer
    finally:
        encoder_stack.FLAGS = old
        os.environ.pop("EDGEDICT_WSR_DELAY", None)
        os.environ.pop("EDGEDICT_WSR_PERSIST", None)
    # the persistent form (default) is ONE launch that carries every frame
    steps, enq, n, slots = encoder_stack.schedule(401, 240, 1024, [1, 2, 1, 1, 1, 1], B=64, chunk=12,
                                                  flags=encoder_stack.WSR)
    assert n == 1 and slots == 6 and all((s == 0).all() for s in steps)
