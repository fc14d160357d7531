"""CPU: the front-end oracle (oracle/frontend_ref.py) against the reference's own FrontEnd outputs
(tests/golden/frontend.npz, oracle/make_golden_frontend.py), and the frame-count rule."""
import os

import numpy as np
import pytest
import torch

from oracle import frontend_ref as FR
from oracle.make_golden_frontend import CASES, make_wave

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "frontend.npz"))


def seeded_sd(params, seed):
    """Same draws as make_golden_frontend.seeded_state_dict, without the reference module: keys in
    the reference's state-dict order (conv1, encode.{i}.conv, encode.{i}.gn, layer_norm)."""
    shapes = [("conv1.weight", (params[0][2], 1, params[0][0])), ("conv1.bias", (params[0][2],))]
    for i in range(1, len(params)):
        cin, (k, _, cout) = params[i - 1][2], params[i]
        shapes += [("encode.%d.conv.weight" % (i - 1), (cout, cin, k)), ("encode.%d.conv.bias" % (i - 1), (cout,)),
                   ("encode.%d.gn.weight" % (i - 1), (cin,)), ("encode.%d.gn.bias" % (i - 1), (cin,))]
    shapes += [("layer_norm.weight", (params[-1][2],)), ("layer_norm.bias", (params[-1][2],))]
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in shapes:
        if k.endswith("gn.weight") or k == "layer_norm.weight":
            sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            sd[k] = 0.1 * torch.randn(shp, generator=g)
        else:
            sd[k] = torch.randn(shp, generator=g) * (2.0 / (shp[1] * shp[2])) ** 0.5
    return sd


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_reproduces_reference_frontend(name):
    params, B, N, seed = CASES[name]
    sd = seeded_sd(params, seed)
    with torch.no_grad():
        y = FR.frontend_forward(sd, make_wave(B, N, seed), [p[1] for p in params])
    np.testing.assert_allclose(y.numpy(), G[name + "_out"], atol=2e-5)


def test_frame_count_rule():
    # Conv1d(padding k-1, stride s) gives floor((T + k - 2)/s) + 1 frames, the block drops k-1
    for T, k, s in [(4000, 10, 5), (799, 8, 4), (301, 4, 2), (9, 3, 2), (5, 2, 2)]:
        x = torch.zeros(1, 1, T)
        y = FR.causal_conv(x, torch.zeros(1, 1, k), None, s)
        assert y.shape[2] == (T + k - 2) // s + 1 - (k - 1)
