"""GPU: the HIP FrontEnd (csrc/frontend.hip + GEMMs) against the reference's own outputs
(tests/golden/frontend.npz) and, for every parameter gradient, against CPU autograd through the
oracle in float64."""
import os

import numpy as np
import pytest
import torch

from oracle import frontend_ref as FR
from oracle.make_golden_frontend import CASES, make_wave
from test_oracle_frontend import seeded_sd

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "frontend.npz"))


def _engine(params, sd, dtype="fp32"):
    from edgedict_amd.models import FrontEnd
    m = FrontEnd(frontend_params=params, bias=True)
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    m.compute_dtype = dtype
    return m


@pytest.mark.parametrize("name", list(CASES))
def test_frontend_matches_reference_golden(hip_lib, name):
    params, B, N, seed = CASES[name]
    sd = seeded_sd(params, seed)
    m = _engine(params, sd)
    with torch.no_grad():
        y = m(make_wave(B, N, seed).cuda())
    assert y.shape == G[name + "_out"].shape and y.shape[1] == m.out_frames(N)
    np.testing.assert_allclose(y.float().cpu().numpy(), G[name + "_out"], atol=1e-4)
    with torch.no_grad():     # B x 1 x T input form (rnnt/models.py:352-353)
        y3 = m(make_wave(B, N, seed).cuda().unsqueeze(1))
    assert torch.equal(y3, y)


@pytest.mark.parametrize("name", ["small", "default"])
def test_frontend_gradients_match_cpu_autograd(hip_lib, name):
    params, B, N, seed = CASES[name]
    sd = seeded_sd(params, seed)
    x = make_wave(B, N, seed)
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    y64 = FR.frontend_forward(sd64, x.double(), [p[1] for p in params])
    w = torch.randn(y64.shape, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    (y64 * w).sum().backward()
    m = _engine(params, sd)
    y = m(x.cuda())
    (y * w.float().cuda()).sum().backward()
    for k, p in m.named_parameters():
        ref = sd64[k].grad
        assert p.grad is not None, k
        scale = max(ref.abs().max().item(), 1e-6)
        err = (p.grad.double().cpu() - ref).abs().max().item() / scale
        assert err < 2e-3, (k, err)


def test_frontend_bf16_tracks_fp32_and_feeds_the_transducer(hip_lib):
    """cli/train.py:107-125: FrontEnd(...) -> Transducer(input_size=128, enc_time_reductions=[])."""
    from edgedict_amd.models import Transducer
    params, B, N, seed = CASES["default"]
    sd = seeded_sd(params, seed)
    x = make_wave(B, N, seed).cuda()
    y32 = _engine(params, sd)(x)
    fe = _engine(params, sd, "bf16")
    y16 = fe(x)
    assert y16.dtype == torch.bfloat16
    assert (y16.float() - y32.float()).abs().max().item() < 0.15
    t = Transducer(16, 40, 128, 32, 2, 0.0, 24, 32, 1, 0.0, 24, 32, enc_time_reductions=[]).cuda()
    t.compute_dtype = "bf16"
    ys = torch.randint(4, 40, (B, 3), dtype=torch.int32).cuda()
    T = y16.shape[1]
    loss = t(y16, ys, torch.full((B,), T, dtype=torch.int32), torch.full((B,), 3, dtype=torch.int32))
    loss.backward()
    assert torch.isfinite(loss).all()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in fe.parameters())
