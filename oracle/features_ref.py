"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement of the reference's log-mel front-end:
    FilterbankFeatures.__init__/forward   rnnt/features.py:40-152  (twin: parts/features.py:228-357)
    Downsample.forward                    rnnt/transforms.py:38-51
The reference module itself cannot be imported here (needs librosa, and its legacy
``torch.stft`` call without ``return_complex`` raises on torch 2.x), so this file follows the
cited lines with ``torch.stft(..., return_complex=True)`` and an independent restatement of
``librosa.filters.mel`` (0.7.2 defaults: Slaney scale, area normalisation).

PARITY STATUS: **unpinned by reference outputs** (the reference holds no fixture for this path
and cannot run here); pinned only on closed-form properties checked in
tests/test_oracle_features.py (frame counts of SURVEY.md A1, filter normalisation, known mel
edge frequencies).
"""
import math

import numpy as np
import torch


def hz_to_mel(f):
    f = float(f)
    if f < 1000.0:
        return f * 3.0 / 200.0
    return 15.0 + math.log(f / 1000.0) * 27.0 / math.log(6.4)


def mel_to_hz(m):
    m = float(m)
    if m < 15.0:
        return m * 200.0 / 3.0
    return 1000.0 * math.exp((m - 15.0) * math.log(6.4) / 27.0)


def mel_filters(sr=16000, n_fft=512, n_mels=80, fmin=0.0, fmax=None):
    fmax = sr / 2.0 if fmax is None else fmax
    lo, hi = hz_to_mel(fmin), hz_to_mel(fmax)
    edges = [mel_to_hz(lo + (hi - lo) * i / (n_mels + 1)) for i in range(n_mels + 2)]
    nb = n_fft // 2 + 1
    fb = np.zeros((n_mels, nb), dtype=np.float64)
    for m in range(n_mels):
        e0, e1, e2 = edges[m], edges[m + 1], edges[m + 2]
        for k in range(nb):
            fk = k * (sr / 2.0) / (nb - 1)
            up = (fk - e0) / (e1 - e0)
            down = (e2 - fk) / (e2 - e1)
            fb[m, k] = max(0.0, min(up, down)) * 2.0 / (e2 - e0)
    return fb.astype(np.float32)


def log_fbank(x, win_length=320, hop_length=160, n_fft=512, n_filt=80, preemph=0.97,
              sample_rate=16000, log=True):
    """x: float32 [B, N] (dither off) -> [B, n_filt, 1 + N // hop]."""
    B, N = x.shape
    seq_len = math.ceil(N / hop_length)
    if preemph is not None:
        x = torch.cat([x[:, :1], x[:, 1:] - preemph * x[:, :-1]], dim=1)
    window = torch.hann_window(win_length, periodic=False, dtype=torch.float32)
    spec = torch.stft(x, n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window,
                      center=True, pad_mode="reflect", return_complex=True)
    power = spec.real ** 2 + spec.imag ** 2
    fb = torch.from_numpy(mel_filters(sample_rate, n_fft, n_filt))
    feat = torch.matmul(fb.unsqueeze(0), power)
    if log:
        feat = torch.log(feat + 1e-20)
    mask = torch.arange(feat.shape[-1]) >= seq_len
    return feat.masked_fill(mask[None, None, :], 0.0)


def downsample(feat, n_frame=3, pad_to_divisible=True):
    """[B, n_filt, F] -> [B, n_filt*n_frame, T0] as rnnt/transforms.py:38-51."""
    feat = feat.transpose(1, 2)
    B, L, D = feat.shape
    if pad_to_divisible:
        pad = (n_frame - L % n_frame) % n_frame
        feat = torch.cat([feat, feat.new_zeros(B, pad, D)], dim=1)
    else:
        feat = feat[:, :L - L % n_frame]
    return feat.reshape(B, -1, D * n_frame).transpose(1, 2)


def stacked_features(x, n_frame=3, pad_to_divisible=True, **kw):
    """waveform [B,N] -> model input [B, T0, n_filt*n_frame] (dataset transposes, rnnt/dataset.py:103)."""
    return downsample(log_fbank(x, **kw), n_frame, pad_to_divisible).transpose(1, 2)
