#!/usr/bin/env python
"""Generate tests/golden/stream.npz by EXECUTING the reference's own streaming decoder loop,
``PytorchStreamDecoder.reset`` / ``.decode`` (rnnt/stream.py:78-120), lifted from its file with
``ast`` (the module cannot be imported here: torchaudio / absl are absent).  ``__init__`` (:29-76:
flag parsing, checkpoint and BPE-vocabulary loading) is bypassed; the attributes it would set are
filled with

  * ``encoder`` / ``decoder`` / ``joint``: sub-modules of the REFERENCE ``rnnt.models.Transducer``
    (imported from /root/reference, as oracle/make_golden.py does) holding seeded weights,
  * ``transform``: what ``build_transform(..., pad_to_divisible=False)[1]`` builds for
    feature='logfbank' (rnnt/transforms.py:165-203): the reference's FilterbankFeatures followed by
    its Downsample(3, False), both lifted by oracle/make_golden_features.py - with dither switched
    off (the reference leaves the class default 1e-5 on: random per chunk, not reproducible),
  * ``tokenizer.tokenizer.id_to_token``: a stub vocabulary ('<unk>' for id 3, 't<id></w>' otherwise),
    so the returned text encodes the emitted ids.

Only the texts the reference returns per chunk are stored, for single streams and for S independent
streams with resets (the fixture of the batched decoder).  Asserts oracle/stream_ref.py reproduces them.

    python oracle/make_golden_stream.py        # needs /root/reference
"""
import ast
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import models_ref as M                       # noqa: E402
from oracle.make_golden import reference_model            # noqa: E402
from oracle import make_golden_features as GF             # noqa: E402

SMALL = dict(vocab_embed_size=16, vocab_size=64, input_size=240, enc_hidden_size=64, enc_layers=3,
             enc_proj_size=48, dec_hidden_size=32, dec_layers=2, dec_proj_size=32, joint_size=64)
E6D2 = dict(vocab_embed_size=64, vocab_size=2048, input_size=240, enc_hidden_size=1024,
            enc_layers=6, enc_proj_size=640, dec_hidden_size=256, dec_layers=2,
            dec_proj_size=256, joint_size=640)

# name: (cfg, weight seed, wave seed, streams, chunks, resets {chunk: [streams]}, (blank bias, <unk> bias))
# the biases are tuned (oracle scan) so that blanks, ordinary symbols and the '<unk>' rule of
# rnnt/stream.py:105-108 all occur: with +2.05 / +2.1 the first arg-max is '<unk>' on every frame and
# the re-arg-max after zeroing that logit picks blank or a symbol; with +1.9 only on some frames
CASES = {
    "small": (SMALL, 5, 0, 1, 11, {}, (0.8, 2.05)),
    "small_multi": (SMALL, 5, 1, 5, 6, {3: [1, 4]}, (0.8, 1.9)),
    "E6D2": (E6D2, 31, 2, 1, 16, {8: [0]}, (0.55, 2.1)),
    "E6D2_multi": (E6D2, 31, 3, 3, 8, {}, (0.55, 1.9)),
}
WIN, HOP = 1320, 1200      # stream.py:71-77 at E6D2 framing: win 320, hop 200, downsample 3, 2 frames


def state_dict(cfg, seed, bias=(0.8, 2.05)):
    """Seeded weights with the joint's output bias raised for blank (id 0) and '<unk>' (id 3)."""
    sd = M.make_state_dict(cfg, seed)
    sd["joint.joint.2.bias"][0] += bias[0]
    sd["joint.joint.2.bias"][3] += bias[1]
    return sd


class StubVocab:
    def id_to_token(self, i):
        return {0: "<nul>", 1: "<pad>", 2: "<bos>", 3: "<unk>"}.get(i, "t%d</w>" % i)


def reference_decoder_class():
    path = os.path.join(REF, "rnnt", "stream.py")
    tree = ast.parse(open(path).read())
    keep = [n for n in tree.body if isinstance(n, ast.ClassDef)
            and n.name in ("StreamTransducerDecoder", "PytorchStreamDecoder")]
    assert len(keep) == 2
    ns = {"torch": torch, "time": time, "os": os, "np": np, "BOS": M.BOS, "NUL": M.NUL}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns["PytorchStreamDecoder"]


def reference_stream(cfg, sd, Decoder, RF, DS):
    m = reference_model(cfg, sd)
    d = object.__new__(Decoder)                      # __init__ bypassed, see the module docstring
    d.FLAGS = types.SimpleNamespace(enc_layers=cfg["enc_layers"], enc_hidden_size=cfg["enc_hidden_size"],
                                    dec_layers=cfg["dec_layers"], dec_hidden_size=cfg["dec_hidden_size"])
    d.tokenizer = types.SimpleNamespace(tokenizer=StubVocab())
    fb = RF(n_filt=80, n_fft=512, win_length=320, hop_length=200)
    fb.dither = 0.0
    d.transform = torch.nn.Sequential(fb, DS(3, False))
    d.encoder, d.decoder, d.joint = m.encoder, m.decoder, m.joint
    d.reset_profile()
    d.reset()
    return d


def ids_of(text):
    return [int(t[1:]) for t in text.split(" ") if t]


def main():
    from oracle.stream_ref import StreamOracle
    from edgedict_amd.flags import make_flags
    torch.set_num_threads(8)
    Decoder = reference_decoder_class()
    RF, DS = GF.reference_rnnt_features(), GF.reference_downsample()
    flags = make_flags("E6D2")
    out = {}
    for name, (cfg, wseed, xseed, S, n_chunks, resets, bias) in CASES.items():
        sd = state_dict(cfg, wseed, bias)
        g = torch.Generator(device="cpu").manual_seed(xseed)
        wave = 0.1 * torch.randn(S, WIN + n_chunks * HOP, generator=g)
        refs = [reference_stream(cfg, sd, Decoder, RF, DS) for _ in range(S)]
        oracles = [StreamOracle(sd, flags) for _ in range(S)]
        texts = np.empty((S, n_chunks), dtype=object)
        emitted = blanks = 0
        for c in range(n_chunks):
            for s in resets.get(c, []):
                refs[s].reset()
                oracles[s].reset()
            for s in range(S):
                chunk = wave[s:s + 1, c * HOP:c * HOP + WIN]
                text = refs[s].decode(chunk.clone())
                full = oracles[s].decode(chunk.clone())
                want = [t for t in full if t != 0]
                assert ids_of(text) == want, (name, c, s, text, want)
                texts[s, c] = text
                emitted += len(want)
                blanks += len(full) - len(want)
        assert emitted > 0 and blanks > 0, name
        assert len(refs[0].encoder_elapsed) == n_chunks
        out[name + "_texts"] = texts.astype(str)
        out[name + "_cfg"] = np.array([wseed, xseed, S, n_chunks])
        print("%s: %d streams x %d chunks, %d symbols + %d blanks, oracle == reference" % (name, S, n_chunks, emitted, blanks))
    path = os.path.join(ROOT, "tests", "golden", "stream.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
