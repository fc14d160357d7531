"""TEST INFRASTRUCTURE ONLY (never imported by the product path).

CPU restatement of the reference's batch assembly, rnnt/dataset.py:202-240, in plain loops:
  zero_pad_concat (202-211): f32 zeros [B, max_i T_i, ...], row i gets feats[i] in its first T_i
  end_pad_concat  (214-222): int64 [B, max_i U_i] filled with PAD (= 1, rnnt/tokenizer.py), row i
                             gets texts[i]
  seq_collate     (225-240): xs as above, ys = end_pad_concat(...).int(), xlen / ylen int32
Pinned on tests/golden/collate.npz, produced by the reference's own functions
(oracle/make_golden_collate.py).
"""
import numpy as np

PAD = 1


def zero_pad_concat(feats):
    max_t = 0
    for f in feats:
        max_t = max(max_t, f.shape[0])
    out = np.zeros((len(feats), max_t) + tuple(feats[0].shape[1:]), dtype=np.float32)
    for i, f in enumerate(feats):
        for t in range(f.shape[0]):
            out[i, t] = f[t]
    return out


def end_pad_concat(texts):
    max_u = 0
    for t in texts:
        max_u = max(max_u, len(t))
    out = np.full((len(texts), max_u), PAD, dtype=np.int64)
    for i, t in enumerate(texts):
        for u in range(len(t)):
            out[i, u] = t[u]
    return out


def seq_collate(results):
    xs = zero_pad_concat([np.asarray(r[0], dtype=np.float32) for r in results])
    ys = end_pad_concat([np.asarray(r[1]) for r in results]).astype(np.int32)
    xlen = np.array([len(r[0]) for r in results], dtype=np.int32)
    ylen = np.array([len(r[1]) for r in results], dtype=np.int32)
    return xs, ys, xlen, ylen
