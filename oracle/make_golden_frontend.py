#!/usr/bin/env python
"""Generate tests/golden/frontend.npz by running the REFERENCE's own rnnt.models.FrontEnd
(rnnt/models.py:341-365) on seeded weights and waveforms, and assert that oracle/frontend_ref.py
reproduces it.  Build container only (needs /root/reference).

    python oracle/make_golden_frontend.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import frontend_ref as FR        # noqa: E402
from oracle.make_golden import reference_model  # noqa: E402,F401  (loads the reference's rnnt.models)

CASES = {
    # name: (frontend_params, B, N, seed)
    "default": ([(10, 5, 16)] + [(8, 4, 32)] + [(4, 2, 128)] * 3, 2, 4000, 7),     # rnnt/models.py:342
    "train": ([(10, 5, 32)] + [(3, 2, 128)] * 4 + [(2, 2, 128)] * 3, 3, 6000, 8),  # cli/train.py:107-110
    "small": ([(4, 2, 8), (3, 2, 24), (2, 1, 16)], 2, 301, 9),
}


def seeded_state_dict(ref_module, seed):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in ref_module.state_dict().items():
        if k.endswith("gn.weight") or k == "layer_norm.weight":
            sd[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        elif k.endswith("bias"):
            sd[k] = 0.1 * torch.randn(v.shape, generator=g)
        else:
            fan = v.shape[1] * v.shape[2]
            sd[k] = torch.randn(v.shape, generator=g) * (2.0 / fan) ** 0.5
    return sd


def make_wave(B, N, seed):
    g = torch.Generator().manual_seed(seed + 100)
    return (0.1 * torch.randn(B, N, generator=g)).clamp(-1, 1)


def main():
    try:
        reference_model(dict(vocab_embed_size=8, vocab_size=40, input_size=24, enc_hidden_size=32,
                             enc_layers=1, enc_proj_size=24, dec_hidden_size=16, dec_layers=1,
                             dec_proj_size=16, joint_size=32), None)
    except Exception:
        pass     # only needed for its side effect: sys.modules["rnnt.models"] is the reference's
    ref_models = sys.modules["rnnt.models"]
    out = {}
    for name, (params, B, N, seed) in CASES.items():
        ref = ref_models.FrontEnd(frontend_params=params, bias=True).eval()
        sd = seeded_state_dict(ref, seed)
        ref.load_state_dict(sd, strict=True)
        x = make_wave(B, N, seed)
        with torch.no_grad():
            y = ref(x)
            o = FR.frontend_forward(sd, x, [p[1] for p in params])
        err = (y - o).abs().max().item()
        assert err < 2e-5, (name, err)
        out[name + "_out"] = y.numpy()
        print("%s: reference FrontEnd %s -> %s, oracle max |d| = %.2e" % (name, tuple(x.shape), tuple(y.shape), err))
    path = os.path.join(ROOT, "tests", "golden", "frontend.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
