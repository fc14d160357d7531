#!/usr/bin/env python
"""Generate tests/golden/specaug.npz by running the REFERENCE's own TimeMasking / FrequencyMasking
classes (rnnt/transforms.py:54-146).  The module itself cannot be imported here (it needs
torchaudio), so the two class definitions are lifted out of the file with ``ast`` at generation
time and executed; nothing of the reference is copied into the repository — only the resulting
input/output vectors.  Also asserts that oracle/transforms_ref.py reproduces them exactly.

    python oracle/make_golden_specaug.py        # needs /root/reference
"""
import ast
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/rnnt/transforms.py"

CASES = [  # (seed, B, F, T, T_mask, T_num_mask, F_mask, F_num_mask)
    (0, 3, 240, 41, 50, 2, 5, 1),
    (1, 4, 240, 167, 50, 2, 5, 1),
    (2, 2, 24, 9, 4, 3, 7, 2),
]


def reference_classes():
    src = open(REF).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in ("TimeMasking", "FrequencyMasking")]
    assert len(keep) == 2
    mod = ast.Module(body=keep, type_ignores=[])
    ns = {"torch": torch, "random": random}
    exec(compile(mod, REF, "exec"), ns)
    return ns["TimeMasking"], ns["FrequencyMasking"]


def main():
    from oracle import transforms_ref as Tr
    TimeMasking, FrequencyMasking = reference_classes()
    out = {}
    for ci, (seed, B, F, T, tm, tn, fm, fn) in enumerate(CASES):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(B, F, T, generator=g) + 3.0     # no exact zeros in the input
        random.seed(seed)
        y = FrequencyMasking(fm, fn)(TimeMasking(tm, tn)(x))     # Sequential order of build_transform
        random.seed(seed)
        mine = Tr.spec_augment(x, tm, tn, fm, fn)
        assert torch.equal(y, mine), "oracle restatement differs from the reference"
        # x is regenerated from the seed by the tests; the fixture keeps the reference's zero mask
        assert torch.equal(y, x.masked_fill(y == 0, 0))
        out["mask%d" % ci] = np.packbits((y == 0).numpy())
        out["cfg%d" % ci] = np.array([seed, B, F, T, tm, tn, fm, fn])
    path = os.path.join(ROOT, "tests", "golden", "specaug.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
