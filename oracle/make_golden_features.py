#!/usr/bin/env python
"""Generate tests/golden/features.npz by EXECUTING the reference's own feature code:

  * ``FilterbankFeatures`` of rnnt/features.py:33-152 (the whole module is lifted with ``ast``;
    only its ``import`` statements are dropped),
  * ``Downsample`` of rnnt/transforms.py:30-51,
  * the Jasper-derived twin ``FilterbankFeatures`` of parts/features.py:228-357 together with its
    helpers ``normalize_batch`` (:80-109) and ``splice_frames`` (:111-123; the ``@torch.jit.script``
    decorator is dropped - exec'd code has no retrievable source - which does not change the
    arithmetic).

The modules cannot be IMPORTED in this environment (librosa / torchaudio are absent and the legacy
``torch.stft`` call without ``return_complex`` raises on torch 2.x), so the lifted code runs in a
namespace whose ``torch`` is a thin proxy that only rewrites ``stft`` to the modern spelling
(``return_complex=True`` + ``view_as_real``: the same numbers in the legacy [.., 2] layout), and
whose ``librosa.filters.mel`` is HuggingFace ``transformers.audio_utils.mel_filter_bank(norm='slaney',
mel_scale='slaney')`` - an INDEPENDENT third-party implementation of the librosa filter bank (it is
what Whisper's feature extractor uses in place of librosa).  Nothing of the reference is written
into the repository - only input seeds and output vectors.  The script also asserts that
``oracle/features_ref.py`` reproduces every output.

    python oracle/make_golden_features.py        # needs /root/reference and transformers
"""
import ast
import math
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"


class _TorchProxy:
    """``torch`` for the lifted code: everything is the real module except the legacy stft call."""

    def __getattr__(self, name):
        return getattr(torch, name)

    @staticmethod
    def stft(x, **kw):
        return torch.view_as_real(torch.stft(x, return_complex=True, **kw))


def third_party_mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) (0.7.2 defaults: htk=False, norm=1) through
    transformers' implementation; returns [n_mels, 1 + n_fft // 2] float32 like librosa."""
    from transformers.audio_utils import mel_filter_bank
    fb = mel_filter_bank(num_frequency_bins=1 + n_fft // 2, num_mel_filters=n_mels,
                         min_frequency=float(fmin), max_frequency=float(fmax if fmax else sr / 2),
                         sampling_rate=sr, norm="slaney", mel_scale="slaney")
    return np.ascontiguousarray(fb.T).astype(np.float32)


def _lift(path, keep=None, strip_decorators=False):
    tree = ast.parse(open(path).read())
    body = []
    for n in tree.body:
        if isinstance(n, (ast.Import, ast.ImportFrom)):
            continue
        if keep is not None:
            names = [n.name] if isinstance(n, (ast.ClassDef, ast.FunctionDef)) else \
                [t.id for t in getattr(n, "targets", []) if isinstance(t, ast.Name)]
            if not any(x in keep for x in names):
                continue
        if strip_decorators and isinstance(n, ast.FunctionDef):
            n.decorator_list = []
        body.append(n)
    librosa = types.SimpleNamespace(filters=types.SimpleNamespace(mel=third_party_mel))
    proxy = _TorchProxy()
    ns = {"torch": proxy, "nn": torch.nn, "math": math, "librosa": librosa}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


def reference_rnnt_features():
    return _lift(os.path.join(REF, "rnnt", "features.py"))["FilterbankFeatures"]


def reference_downsample():
    return _lift(os.path.join(REF, "rnnt", "transforms.py"), keep={"Downsample"})["Downsample"]


def reference_parts_features():
    ns = _lift(os.path.join(REF, "parts", "features.py"),
               keep={"normalize_batch", "splice_frames", "constant", "FilterbankFeatures"},
               strip_decorators=True)
    return ns["FilterbankFeatures"]


def wave(seed, B, N):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return 0.1 * torch.randn(B, N, generator=g)


# rnnt/features.py cases: (seed, B, N, win_length, hop_length, n_filt, sample stride over frames)
RNNT_CASES = [
    (11, 2, 8000, 320, 160, 80, 1),      # E4D1 geometry, hop | N -> last frame zeroed
    (12, 2, 8123, 320, 200, 80, 1),      # E6D2 geometry, ragged tail
    (13, 1, 12000, 400, 320, 80, 1),     # E6D2_LARGE_Batch geometry (win 400, hop 320)
    (14, 3, 1320, 320, 200, 80, 1),      # one streaming chunk (stream.py: win_size 1320)
    (15, 1, 240000, 320, 200, 80, 25),   # full 15 s utterance, every 25th frame kept
    (16, 2, 4000, 320, 160, 64, 1),      # the class default n_filt
]

# parts/features.py cases: ctor kwargs + (seed, B, N, seq_len list)
PARTS_CASES = [
    (dict(sample_rate=16000, window_size=0.02, window_stride=0.01, window="hamming",
          normalize="per_feature", nfilt=64, dither=0.0, pad_to=8, frame_splicing=1),
     21, 3, 6400, [6400, 5000, 3210]),
    (dict(sample_rate=16000, window_size=0.02, window_stride=0.0125, window="hann",
          normalize="none", n_fft=512, nfilt=80, dither=0.0, pad_to=0, frame_splicing=3),
     22, 2, 8123, [8123, 4000]),
    (dict(sample_rate=16000, window_size=0.025, window_stride=0.01, window="blackman",
          normalize="all_features", nfilt=40, dither=0.0, pad_to=16, frame_splicing=1, log=True),
     23, 2, 4800, [4800, 4321]),
    (dict(sample_rate=8000, window_size=0.02, window_stride=0.01, window="bartlett",
          normalize="per_feature", nfilt=64, dither=0.0, pad_to=8, frame_splicing=2, preemph=None),
     24, 2, 3000, [3000, 1500]),
    # short input: fewer samples than n_fft -> zero-padded to win_length (parts/features.py:289-294)
    (dict(sample_rate=16000, window_size=0.02, window_stride=0.01, window="hamming",
          normalize="none", nfilt=64, dither=0.0, pad_to=0, frame_splicing=1),
     25, 1, 300, [300]),
]


def main():
    from oracle import features_ref as Fr
    RF = reference_rnnt_features()
    DS = reference_downsample()
    PF = reference_parts_features()
    out = {"mel_80": third_party_mel(16000, 512, 80, 0, 8000), "mel_64": third_party_mel(16000, 512, 64, 0, 8000),
           "mel_8k_64": third_party_mel(8000, 256, 64, 0, 4000)}
    worst = 0.0
    with torch.no_grad():
        for i, (seed, B, N, win, hop, nf, stride) in enumerate(RNNT_CASES):
            x = wave(seed, B, N)
            m = RF(sample_rate=16000, win_length=win, hop_length=hop, n_fft=512, dither=0.0, n_filt=nf)
            y = m(x.clone())
            mine = Fr.log_fbank(x, win_length=win, hop_length=hop, n_fft=512, n_filt=nf)
            assert y.shape == mine.shape, (y.shape, mine.shape)
            worst = max(worst, (y - mine).abs().max().item())
            for pad in (True, False):
                z = DS(3, pad)(y)
                zm = Fr.downsample(mine, 3, pad)
                assert z.shape == zm.shape
                assert torch.equal(z, DS(3, pad)(y)) and (z - zm).abs().max().item() <= worst
                if stride == 1:
                    out["rnnt%d_stack_%s" % (i, "pad" if pad else "trunc")] = z.numpy()
            out["rnnt%d_cfg" % i] = np.array([seed, B, N, win, hop, nf, stride])
            out["rnnt%d_feat" % i] = y.numpy()[:, :, ::stride]
        for i, (kw, seed, B, N, seq) in enumerate(PARTS_CASES):
            x = wave(seed, B, N)
            sl = torch.tensor(seq, dtype=torch.int32)
            m = PF(**kw)
            y = m(x.clone(), sl)
            mine = Fr.parts_log_fbank(x, sl, **kw)
            assert y.shape == mine.shape, (i, y.shape, mine.shape)
            err = (y - mine).abs().max().item()
            worst = max(worst, err)
            out["parts%d_feat" % i] = y.numpy()
            out["parts%d_cfg" % i] = np.array([seed, B, N] + seq)
    assert worst < 5e-5, worst
    path = os.path.join(ROOT, "tests", "golden", "features.npz")
    np.savez_compressed(path, **out)
    print("reference-executed features vs oracle restatement: max |diff| = %.2e; wrote %s (%d KB)"
          % (worst, path, os.path.getsize(path) // 1024))


if __name__ == "__main__":
    main()
