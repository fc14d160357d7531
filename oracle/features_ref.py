"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement of the reference's log-mel front-end:
    FilterbankFeatures.__init__/forward   rnnt/features.py:40-152  (twin: parts/features.py:228-357)
    Downsample.forward                    rnnt/transforms.py:38-51
The reference module itself cannot be imported here (needs librosa, and its legacy
``torch.stft`` call without ``return_complex`` raises on torch 2.x), so this file follows the
cited lines with ``torch.stft(..., return_complex=True)`` and an independent restatement of
``librosa.filters.mel`` (0.7.2 defaults: Slaney scale, area normalisation).

PARITY STATUS: **pinned**.  ``oracle/make_golden_features.py`` EXECUTES the reference's own
``FilterbankFeatures`` (rnnt/features.py and the parts/features.py twin) and ``Downsample`` classes,
lifted from their files with ``ast`` (torch.stft respelled for torch 2.x, ``librosa.filters.mel``
supplied by HuggingFace transformers' independent Slaney implementation), stores their outputs in
``tests/golden/features.npz`` and asserts this restatement reproduces them;
``tests/test_oracle_features.py`` re-checks it on every run without the reference.
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


# ---------------------------------------------------------------- parts/features.py twin
_WINDOWS = {"hann": torch.hann_window, "hamming": torch.hamming_window,
            "blackman": torch.blackman_window, "bartlett": torch.bartlett_window}


def normalize_batch(x, seq_len, normalize_type):
    """parts/features.py:80-109 (rnnt/features.py:7-30): per-utterance mean / UNBIASED std over the
    first seq_len frames, per feature row or over all features; std += 1e-5."""
    if normalize_type == "per_feature":
        out = torch.empty_like(x)
        for i in range(x.shape[0]):
            n = int(seq_len[i])
            mean = x[i, :, :n].mean(dim=1)
            std = x[i, :, :n].std(dim=1) + 1e-5
            out[i] = (x[i] - mean[:, None]) / std[:, None]
        return out
    if normalize_type == "all_features":
        out = torch.empty_like(x)
        for i in range(x.shape[0]):
            n = int(seq_len[i])
            out[i] = (x[i] - x[i, :, :n].mean()) / (x[i, :, :n].std() + 1e-5)
        return out
    return x


def parts_log_fbank(x, seq_len, sample_rate=8000, window_size=0.02, window_stride=0.01,
                    window="hamming", normalize="per_feature", n_fft=None, preemph=0.97, nfilt=64,
                    lowfreq=0, highfreq=None, log=True, dither=0.0, pad_to=8, max_duration=16.7,
                    frame_splicing=1):
    """Restatement of the Jasper-derived twin, parts/features.py:232-347 (dither off):
    seconds-based geometry (:250-252), inputs shorter than n_fft zero-padded to win_length
    (:289-294), ``forward(x, seq_len)`` with per-utterance sample counts (:298-301), frame
    "splicing" (:111-123 - as written it concatenates ``frame_splicing`` COPIES of the features
    along the feature axis: cat(x[:, :, :n+1], x[:, :, n+1:]) is x), normalisation (:329), masking
    beyond ceil(seq_len / hop) (:332-336) and padding of the frame axis (:339-345: with pad_to > 0
    ALWAYS pad_to - F % pad_to extra frames, i.e. a full pad_to when F is already a multiple)."""
    assert dither == 0.0
    win_length = int(sample_rate * window_size)
    hop_length = int(sample_rate * window_stride)
    n_fft = n_fft or 2 ** math.ceil(math.log2(win_length))
    highfreq = highfreq or sample_rate / 2
    frames = torch.ceil(seq_len.float() / hop_length).int()
    if preemph is not None:
        x = torch.cat([x[:, :1], x[:, 1:] - preemph * x[:, :-1]], dim=1)
    if x.shape[-1] < n_fft:
        if x.shape[-1] > win_length:
            raise RuntimeError("parts FilterbankFeatures: input longer than win_length but shorter "
                               "than n_fft cannot be padded (the reference's copy fails the same way)")
        x = torch.cat([x, x.new_zeros(x.shape[0], win_length - x.shape[-1])], dim=1)
    win = _WINDOWS[window](win_length, periodic=False, dtype=torch.float32) if window in _WINDOWS else None
    spec = torch.stft(x, n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=win,
                      center=True, pad_mode="reflect", return_complex=True)
    power = spec.real ** 2 + spec.imag ** 2
    fb = torch.from_numpy(mel_filters(sample_rate, n_fft, nfilt, lowfreq, highfreq))
    feat = torch.matmul(fb.unsqueeze(0), power)
    if log:
        feat = torch.log(feat + 1e-20)
    if frame_splicing > 1:
        feat = torch.cat([feat] * frame_splicing, dim=1)
    feat = normalize_batch(feat, frames, normalize)
    F_ = feat.shape[-1]
    mask = torch.arange(F_)[None, :] >= frames[:, None]
    feat = feat.masked_fill(mask[:, None, :], 0.0)
    if pad_to < 0:
        max_length = 1 + math.ceil((max_duration * sample_rate - win_length) / hop_length)
        max_length += 16 - (max_length % 16)
        feat = torch.nn.functional.pad(feat, (0, max_length - F_))
    elif pad_to > 0:
        feat = torch.nn.functional.pad(feat, (0, pad_to - F_ % pad_to))
    return feat
