#!/usr/bin/env python
"""Generate tests/golden/beam_tiny.npz: a tiny transducer TRAINED for 500 Adam steps (to a loss of about 1:
confident but not yet collapsed onto one path) on three synthetic utterances (a randomly initialised model
always prefers the empty hypothesis, which exercises nothing), its inputs, and - for several beam widths -

  * the outputs of the REFERENCE's own legacy search, EXECUTED: ``Transducer.beam_search``, ``Sequence`` and
    ``log_aplusb`` are lifted out of /root/reference/models.py:121-224 with ``ast`` (the file cannot be
    imported: it needs the absent ``recurrent`` module) and run, unmodified, on a stub ``self`` whose
    ``encoder`` / ``decoder`` / ``joint`` are the sub-modules of the REFERENCE ``rnnt.models.Transducer``
    holding the trained weights (as oracle/make_golden_stream.py does for the stream decoder).  On torch 2.10
    ``autograd.Variable(..., volatile=True)`` only warns.  The one adaptation is the stub's glue between the
    legacy class's own layers and the maintained model's: ``embed`` passes the token id through and
    ``decoder(label, hidden)`` calls the maintained ``Decoder`` - with ``hidden is None`` (the empty
    hypothesis, whose legacy start token is id 1 from a ``None`` state) it runs ``Decoder.forward(empty,
    None)``, i.e. BOS from the zero state, exactly what rnnt/models.py:150-153,247 start from; the leading
    start token is dropped from the returned sequence;
  * the outputs of oracle/beam_ref.py, asserted EQUAL to the above (tokens, number of
    hypothesis expansions; scores to 1e-6 relative: sums of fp32 log-probabilities from the
    reference's modules vs the oracle's functional restatement) - which is what pins that oracle, and through it the HIP search
    (tests/test_beam_gpu.py), on the reference.

    python oracle/make_golden_beam.py              # train + both searches (needs /root/reference)
    python oracle/make_golden_beam.py --pin-only   # keep the stored weights, re-run both searches
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


REF = "/root/reference"


def reference_beam_search():
    """(beam_search function, namespace) lifted from /root/reference/models.py:121-224."""
    import ast
    import math
    import torch.nn.functional as F
    from torch import autograd
    path = os.path.join(REF, "models.py")
    tree = ast.parse(open(path).read())
    fn = None
    keep = []
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == "Transducer":
            fn = next(n for n in node.body if isinstance(n, ast.FunctionDef) and n.name == "beam_search")
        elif isinstance(node, ast.ClassDef) and node.name == "Sequence":
            keep.append(node)
        elif isinstance(node, ast.FunctionDef) and node.name == "log_aplusb":
            keep.append(node)
    assert fn is not None and len(keep) == 2
    ns = {"torch": torch, "F": F, "autograd": autograd, "math": math}
    exec(compile(ast.Module(body=keep + [fn], type_ignores=[]), path, "exec"), ns)
    return ns["beam_search"]


class _LegacySelf:
    """The attributes models.py:121-202 reads from ``self``, backed by the reference's maintained modules."""

    def __init__(self, ref_model, blank, vocab_size):
        self.m = ref_model
        self.blank = blank
        self.vocab_size = vocab_size
        self.expansions = 0
        self.frames = None

    def encoder(self, xs):
        y, hid = self.m.encoder(xs)
        if self.frames is not None:          # an utterance of a batch uses its scale_length frames
            y = y[:, :self.frames]
        return y, hid

    def embed(self, label):
        return label                          # the maintained Decoder embeds the ids itself

    def decoder(self, label, hidden):
        self.expansions += 1
        if hidden is None:                    # empty hypothesis: BOS from the zero state
            return self.m.decoder(torch.empty(1, 0, dtype=torch.long), None)
        return self.m.decoder(label, hidden)

    def joint(self, x, pred):
        return self.m.joint(x, pred)


def reference_search(cfg, sd, xs, xlen, W, prefix=False):
    import warnings
    from oracle.make_golden import reference_model
    m = reference_model(cfg, sd)
    fn = reference_beam_search()
    seqs, scores, total = [], [], 0
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")      # 'volatile was removed and now has no effect'
        Tp = m.encoder(xs)[0].shape[1]
        lens = M.scale_length(Tp, xlen)
        for b in range(xs.shape[0]):
            stub = _LegacySelf(m, M.NUL, cfg["vocab_size"])
            stub.frames = int(lens[b])
            k, neg = fn(stub, xs[b:b + 1], W=W, prefix=prefix)
            assert k[0] == 1                  # the legacy start token
            seqs.append(np.array(k[1:], dtype=np.int64))
            scores.append(float(neg))
            total += stub.expansions
    return seqs, np.array(scores, dtype=np.float64), total


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    path = os.path.join(ROOT, "tests", "golden", "beam_tiny.npz")
    pin_only = "--pin-only" in sys.argv
    if pin_only:
        old = np.load(path)
        sd = {k[3:]: torch.from_numpy(old[k]) for k in old.files if k.startswith("sd/")}
        xs, ys = torch.from_numpy(old["xs"]), torch.from_numpy(old["ys"])
        xlen, ylen = torch.from_numpy(old["xlen"]), torch.from_numpy(old["ylen"])
    else:
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
        rseqs, rscores, rn = reference_search(CFG, sd, xs, xlen, W)
        print("W=%d" % W, [s.tolist() for s in seqs], scores, "expansions", n, "| reference-executed:",
              [s.tolist() for s in rseqs], rscores, rn)
        # the oracle restatement reproduces the reference's own code: tokens, scores, pops
        for a, b in zip(seqs, rseqs):
            assert np.array_equal(a, b), (W, a, b)
        assert (np.abs(scores - rscores) / np.abs(rscores)).max() < 1e-6, (W, scores, rscores)   # fp32 module vs functional arithmetic
        assert n == rn, (W, n, rn)
        if pin_only:                          # and the stored vectors are these
            for b, sq in enumerate(seqs):
                assert np.array_equal(old["W%d_seq%d" % (W, b)], sq)
            assert int(old["W%d_expansions" % W][0]) == n
        for b, s in enumerate(seqs):
            out["W%d_seq%d" % (W, b)] = s
            out["ref_W%d_seq%d" % (W, b)] = rseqs[b]
        out["W%d_score" % W] = scores
        out["W%d_expansions" % W] = np.array([n])
        out["ref_W%d_score" % W] = rscores
        out["ref_W%d_expansions" % W] = np.array([rn])
    # ---- the prefix=True branch (models.py:145-161): the same two searches with the flag set
    for W in WIDTHS[1:]:
        seqs, scores, n = beam_ref.beam_search(sd, xs, xlen, W=W, prefix=True)
        rseqs, rscores, rn = reference_search(CFG, sd, xs, xlen, W, prefix=True)
        plain = [out["W%d_seq%d" % (W, b)].tolist() for b in range(len(seqs))]
        print("prefix W=%d" % W, [s.tolist() for s in seqs], scores, "expansions", n, "| reference-executed:",
              [s.tolist() for s in rseqs], rscores, rn, "| differs from prefix=False:",
              [s.tolist() for s in seqs] != plain or not np.allclose(scores, out["W%d_score" % W], rtol=1e-9))
        for a, b in zip(seqs, rseqs):
            assert np.array_equal(a, b), (W, a, b)
        assert (np.abs(scores - rscores) / np.abs(rscores)).max() < 1e-6, (W, scores, rscores)
        assert n == rn, (W, n, rn)
        for b, sq in enumerate(seqs):
            out["P_W%d_seq%d" % (W, b)] = sq
            out["ref_P_W%d_seq%d" % (W, b)] = rseqs[b]
        out["P_W%d_score" % W] = scores
        out["P_W%d_expansions" % W] = np.array([n])
        out["ref_P_W%d_score" % W] = rscores
        out["ref_P_W%d_expansions" % W] = np.array([rn])
    g, gs = M.greedy_decode(sd, xs, xlen)
    print("greedy", [[int(t) for t in s if t != 0] for s in g], "labels", [y[:n].tolist() for y, n in zip(ys, ylen)])
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
