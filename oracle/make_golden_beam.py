#!/usr/bin/env python
"""Generate tests/golden/beam_tiny.npz: a tiny transducer TRAINED for 500 Adam steps (to a loss of about 1: confident but not yet collapsed onto one path) on three
synthetic utterances (a randomly initialised model always prefers the empty hypothesis under
oracle/beam_ref.py's search, which exercises nothing), its inputs, and the beam-search oracle's
outputs for several beam widths.  Everything runs on the CPU through the oracle's own functions
(oracle/models_ref.py, oracle/rnnt_loss_ref.py); the reference's legacy beam search itself cannot run
on torch 2.x (see oracle/beam_ref.py), so these vectors pin the HIP path to the ORACLE.

    python oracle/make_golden_beam.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import beam_ref, models_ref as M, rnnt_loss_ref as R  # noqa: E402

CFG = dict(vocab_embed_size=16, vocab_size=40, input_size=24, enc_hidden_size=32, enc_layers=2,
           enc_proj_size=24, dec_hidden_size=32, dec_layers=2, dec_proj_size=24, joint_size=32)
WIDTHS = (1, 2, 4, 10)


def main():
    torch.manual_seed(0)
    sd = M.make_state_dict(CFG, 0)
    xs, ys, xlen, ylen = M.make_batch(CFG, 1, 3, 21, 5)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(list(params.values()), lr=1e-3)
    for step in range(500):
        logits, act = M.transducer_logits(params, xs, ys, xlen, ylen)
        loss = R.rnnt_loss_torch(logits, ys[:, :int(ylen.max())], act, ylen).mean()
        opt.zero_grad()
        loss.backward()
        with torch.no_grad():     # padding_idx row of the embedding stays zero (rnnt/models.py:144)
            params["decoder.embed.weight"].grad[M.PAD] = 0
        opt.step()
        if step % 50 == 0 or step == 499:
            print("step %3d loss %.4f" % (step, float(loss.detach())))
    sd = {k: v.detach().clone() for k, v in params.items()}
    out = {"xs": xs.numpy(), "ys": ys.numpy(), "xlen": xlen.numpy(), "ylen": ylen.numpy()}
    for k, v in sd.items():
        out["sd/" + k] = v.numpy()
    for W in WIDTHS:
        seqs, scores, n = beam_ref.beam_search(sd, xs, xlen, W=W)
        print("W=%d" % W, [s.tolist() for s in seqs], scores, "expansions", n)
        for b, s in enumerate(seqs):
            out["W%d_seq%d" % (W, b)] = s
        out["W%d_score" % W] = scores
        out["W%d_expansions" % W] = np.array([n])
    g, gs = M.greedy_decode(sd, xs, xlen)
    print("greedy", [[int(t) for t in s if t != 0] for s in g], "labels", [y[:n].tolist() for y, n in zip(ys, ylen)])
    path = os.path.join(ROOT, "tests", "golden", "beam_tiny.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
