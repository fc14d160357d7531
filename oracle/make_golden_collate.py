#!/usr/bin/env python
"""Generate tests/golden/collate.npz by running the REFERENCE's own zero_pad_concat /
end_pad_concat / seq_collate (rnnt/dataset.py:202-240).  rnnt.dataset cannot be imported here (it
needs torchaudio), so the three function definitions are lifted out of the file with ``ast`` at
generation time and executed; nothing of the reference is copied into the repository - only the
resulting input/output vectors.  Also asserts that oracle/collate_ref.py reproduces them exactly.

    python oracle/make_golden_collate.py        # needs /root/reference
"""
import ast
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/rnnt/dataset.py"

CASES = [  # (seed, [T_i], [U_i], F)
    (0, [7, 3, 5], [4, 1, 6], 6),
    (1, [1], [1], 3),
    (2, [167, 150, 120, 100], [20, 18, 15, 10], 240),
    (3, [5, 5, 5], [2, 2, 2], 1),
]


def reference_functions():
    tree = ast.parse(open(REF).read())
    names = ("zero_pad_concat", "end_pad_concat", "seq_collate")
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(keep) == 3
    ns = {"torch": torch, "np": np, "PAD": 1}   # PAD: rnnt/tokenizer.py
    exec(compile(ast.Module(body=keep, type_ignores=[]), REF, "exec"), ns)
    return ns["seq_collate"]


def make_case(seed, Ts, Us, F):
    g = torch.Generator().manual_seed(seed)
    feats = [torch.randn(t, F, generator=g) for t in Ts]
    toks = [torch.randint(4, 2048, (u,), generator=g) for u in Us]
    return feats, toks


def main():
    from oracle import collate_ref
    seq_collate = reference_functions()
    out = {}
    for ci, (seed, Ts, Us, F) in enumerate(CASES):
        feats, toks = make_case(seed, Ts, Us, F)
        xs, ys, xlen, ylen = seq_collate(list(zip(feats, toks)))
        assert ys.dtype == torch.int32 and xlen.dtype == torch.int32 and xs.dtype == torch.float32
        rx, ry, rxl, ryl = collate_ref.seq_collate([(f.numpy(), t.numpy()) for f, t in zip(feats, toks)])
        assert np.array_equal(rx, xs.numpy()) and np.array_equal(ry, ys.numpy())
        assert np.array_equal(rxl, xlen.numpy()) and np.array_equal(ryl, ylen.numpy())
        out["c%d_meta" % ci] = np.array([seed, F] + Ts + Us, dtype=np.int64)
        out["c%d_n" % ci] = np.array([len(Ts)], dtype=np.int64)
        out["c%d_xs" % ci] = xs.numpy()
        out["c%d_ys" % ci] = ys.numpy()
        out["c%d_xlen" % ci] = xlen.numpy()
        out["c%d_ylen" % ci] = ylen.numpy()
    path = os.path.join(ROOT, "tests", "golden", "collate.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
