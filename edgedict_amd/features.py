"""Log-mel filterbank features on the GPU (mirror of the reference's ``rnnt/features.py``).

``FilterbankFeatures`` keeps the reference's constructor arguments, buffer names (``fb``,
``window``) and output layout ``[B, n_filt, frames]`` (rnnt/features.py:33-152).
``StackedLogFbank`` is the fused form the engine's trainer uses: filterbank + the frame
stacking of ``Downsample`` (rnnt/transforms.py:30-51) written directly as ``[B, T0, n_filt*k]``.
All arithmetic runs in csrc/fbank.hip; numpy is used only to build the constant tables
(window, FFT twiddles, mel weights) exactly as librosa 0.7.2's ``filters.mel`` defines them.
"""
import math

import numpy as np
import torch
from torch import nn

from ._lib import call, dtype_code, require_cuda
from .ops import _ll


def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mel = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mel)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with its 0.7.2 defaults
    (Slaney scale, ``htk=False``, area normalisation), float32 ``[n_mels, 1+n_fft//2]``."""
    fmax = float(sr) / 2 if fmax is None else fmax
    nbins = 1 + n_fft // 2
    fftfreqs = np.linspace(0, float(sr) / 2, nbins, endpoint=True)
    mel_pts = np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2)
    mel_f = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, nbins), dtype=np.float32)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


def _window_tensor(name, win_length):
    fns = {'hann': torch.hann_window, 'hamming': torch.hamming_window,
           'blackman': torch.blackman_window, 'bartlett': torch.bartlett_window}
    if name not in fns:
        return None
    return fns[name](win_length, periodic=False, dtype=torch.float32)


class FilterbankFeatures(nn.Module):
    """Drop-in for rnnt.features.FilterbankFeatures (normalize='none', pad_to=0 path)."""

    def __init__(self, sample_rate=16000, win_length=320, hop_length=160, n_fft=512,
                 window="hann", normalize="none", log=True, dither=1e-5, pad_to=0,
                 max_duration=16.7, preemph=0.97, n_filt=64, f_min=0, f_max=None):
        super().__init__()
        if normalize not in ("none", None):
            raise NotImplementedError("only normalize='none' (what build_transform uses) is implemented")
        if pad_to != 0:
            raise NotImplementedError("pad_to != 0 is not implemented (the reference never sets it)")
        self.win_length = win_length
        self.hop_length = hop_length
        self.n_fft = n_fft or 2 ** math.ceil(math.log2(win_length))
        self.normalize = normalize
        self.log = log
        self.dither = dither
        self.n_filt = n_filt
        self.preemph = preemph
        self.pad_to = pad_to
        f_max = f_max or sample_rate / 2
        win = _window_tensor(window, win_length)
        if win is None:
            win = torch.ones(win_length)
        fb = mel_filterbank(sample_rate, self.n_fft, n_filt, f_min, f_max)
        self.register_buffer("fb", torch.from_numpy(fb).unsqueeze(0))
        self.register_buffer("window", win)
        # kernel-side constant tables
        lo = (self.n_fft - win_length) // 2
        full = torch.zeros(self.n_fft)
        full[lo:lo + win_length] = win
        k = np.arange(self.n_fft // 2, dtype=np.float64)
        tw = np.stack([np.cos(2 * np.pi * k / self.n_fft), np.sin(2 * np.pi * k / self.n_fft)], 1)
        rng = np.zeros((n_filt, 2), dtype=np.int32)
        for m in range(n_filt):
            nz = np.nonzero(fb[m])[0]
            rng[m] = (nz[0], nz[-1] + 1) if len(nz) else (0, 0)
        self.register_buffer("_window_full", full, persistent=False)
        self.register_buffer("_twiddle", torch.from_numpy(tw.astype(np.float32)), persistent=False)
        self.register_buffer("_fb_range", torch.from_numpy(rng), persistent=False)
        self._win_support = (lo, lo + win_length)
        self._seed = 0
        max_length = 1 + math.ceil((max_duration * sample_rate - win_length) / hop_length)
        self.max_length = max_length + (16 - (max_length % 16))

    def get_seq_len(self, seq_len):
        return torch.ceil(seq_len.float() / self.hop_length).int()

    def n_frames(self, n_samples):
        return 1 + n_samples // self.hop_length

    def _run(self, x, lengths, out, o_b, o_group, o_k, o_m, stack, frames_out):
        if x.is_cuda and self.fb.device != x.device:
            # the window / twiddle / filter tables follow the waveform's device: a caller written for the CPU reference
            # (rnnt/stream.py:38-44 builds the transform and never moves it) only has to hand over device tensors
            self.to(x.device)
        require_cuda(x, self.fb)
        if x.dim() != 2 or x.dtype != torch.float32 or x.stride(1) != 1:
            raise ValueError("waveform must be a float32 [B, N] tensor with unit sample stride")
        B, N = x.shape
        if self.dither > 0:
            # in place on the caller's tensor, as the reference does (rnnt/features.py:111-112)
            self._seed = (self._seed * 1664525 + 1013904223) & 0xFFFFFFFF
            import ctypes
            call("dither", x, _ll(x.stride(0)), B, N, lengths, float(self.dither),
                 ctypes.c_uint(self._seed))
        lo, hi = self._win_support
        call("fbank_forward", x, _ll(x.stride(0)), B, N, lengths, self._window_full, self._twiddle,
             self.fb, self._fb_range, self.n_fft, lo, hi, self.hop_length, self.n_filt,
             float(self.preemph if self.preemph is not None else 0.0), int(bool(self.log)), out,
             dtype_code(out.dtype), _ll(o_b), _ll(o_group), _ll(o_k), _ll(o_m), int(stack),
             int(frames_out))
        return out

    @torch.no_grad()
    def forward(self, x):
        """x: float32 [B, N] -> float32 [B, n_filt, 1 + N // hop] (reference layout)."""
        B, N = x.shape
        F_ = self.n_frames(N)
        out = torch.empty(B, self.n_filt, F_, dtype=torch.float32, device=x.device)
        return self._run(x, None, out, self.n_filt * F_, 1, 0, F_, 1, F_)


class StackedLogFbank(nn.Module):
    """FilterbankFeatures + Downsample(n_frame) + transpose in one kernel:
    waveforms ``[B, N]`` (+ optional per-utterance sample counts) -> ``xs [B, T0, n_filt*k]``
    and ``xlen [B]`` in stacked frames, i.e. exactly what ``seq_collate`` hands to the model
    (rnnt/dataset.py:225-240) when the test transform is logfbank + Downsample."""

    def __init__(self, n_frame=3, pad_to_divisible=True, out_dtype=torch.float32, **fbank_args):
        super().__init__()
        self.fbank = FilterbankFeatures(**fbank_args)
        self.n_frame = n_frame
        self.pad_to_divisible = pad_to_divisible
        self.out_dtype = out_dtype

    def output_frames(self, n_samples):
        F_ = self.fbank.n_frames(n_samples)
        k = self.n_frame
        return (F_ + k - 1) // k if self.pad_to_divisible else F_ // k

    @torch.no_grad()
    def forward(self, x, lengths=None):
        B, N = x.shape
        k, M = self.n_frame, self.fbank.n_filt
        T0 = self.output_frames(N)
        out = torch.empty(B, T0, M * k, dtype=self.out_dtype, device=x.device)
        host_lengths = lengths
        if lengths is not None:
            lengths = lengths.to(device=x.device, dtype=torch.int32, non_blocking=True).contiguous()
        self.fbank._run(x, lengths, out, T0 * M * k, M * k, M, 1, k, T0 * k)
        if lengths is None:
            xlen = torch.full((B,), T0, dtype=torch.int32, device=x.device)
        else:
            # sample counts given on the HOST (what a DataLoader hands over) keep the frame counts
            # on the host too: the model slices by xlen.max() and that must not cost a device sync
            src = host_lengths if not host_lengths.is_cuda else lengths
            F_b = 1 + src.to(torch.int32) // self.fbank.hop_length
            xlen = ((F_b + k - 1) // k if self.pad_to_divisible else F_b // k).to(torch.int32)
        return out, xlen


class PartsFilterbankFeatures(FilterbankFeatures):
    """The Jasper-derived twin of the reference, ``parts/features.py:228-357`` (exported as
    ``parts.features.FilterbankFeatures`` by the import shim): seconds-based constructor,
    ``forward(x, seq_len)`` with per-utterance sample counts, window choice, ``normalize_batch``
    ('per_feature' / 'all_features'), the frame "splicing" of :111-123, inputs shorter than
    ``n_fft`` zero-padded to ``win_length`` (:289-294) and padding of the frame axis (:339-345).
    Same kernel as ``FilterbankFeatures`` (csrc/fbank.hip) through ``edgedict_fbank_forward_masked``:
    here ``seq_len`` only masks frames - the STFT sees the whole padded row, as the reference's does.
    Output: float32 ``[B, nfilt * frame_splicing, frames (+ padding)]``."""

    _NORMALIZE = {"none": 0, None: 0, "per_feature": 1, "all_features": 2}

    def __init__(self, sample_rate=8000, window_size=0.02, window_stride=0.01, window="hamming",
                 normalize="per_feature", n_fft=None, preemph=0.97, nfilt=64, lowfreq=0,
                 highfreq=None, log=True, dither=1e-5, pad_to=8, max_duration=16.7,
                 frame_splicing=1):
        win_length = int(sample_rate * window_size)
        hop_length = int(sample_rate * window_stride)
        super().__init__(sample_rate=sample_rate, win_length=win_length, hop_length=hop_length,
                         n_fft=n_fft, window=window, normalize="none", log=log, dither=dither,
                         pad_to=0, max_duration=max_duration, preemph=preemph, n_filt=nfilt,
                         f_min=lowfreq, f_max=highfreq)
        if normalize not in self._NORMALIZE:
            # the reference's normalize_batch silently returns x for unknown types (:108-109)
            normalize = "none"
        self.normalize = normalize
        self.pad_to = pad_to
        self.nfilt = nfilt
        self.frame_splicing = frame_splicing

    @torch.no_grad()
    def forward(self, x, seq_len):
        require_cuda(x, self.fb)
        if x.dim() != 2 or x.dtype != torch.float32 or x.stride(1) != 1:
            raise ValueError("waveform must be a float32 [B, N] tensor with unit sample stride")
        B, N = x.shape
        n_signal = 0
        if self.dither > 0:      # in place on the caller's tensor (parts/features.py:304-305)
            import ctypes
            self._seed = (self._seed * 1664525 + 1013904223) & 0xFFFFFFFF
            call("dither", x, _ll(x.stride(0)), B, N, None, float(self.dither), ctypes.c_uint(self._seed))
        if N < self.n_fft:       # parts/features.py:289-294
            if N > self.win_length:
                raise RuntimeError("The expanded size of the tensor (%d) must match the existing size "
                                   "(%d): an input longer than win_length but shorter than n_fft cannot "
                                   "be zero-padded (parts/features.py:289-294 fails the same way)"
                                   % (self.win_length, N))
            padded = torch.zeros(B, self.win_length, dtype=torch.float32, device=x.device)
            padded[:, :N] = x
            n_signal = N         # the reference pads AFTER its pre-emphasis: the pad stays exactly zero
            x, N = padded, self.win_length
        F_ = self.n_frames(N)
        rows = self.n_filt * self.frame_splicing
        if self.pad_to < 0:
            Fp = max(self.max_length, F_)
        elif self.pad_to > 0:
            Fp = F_ + self.pad_to - F_ % self.pad_to     # a whole pad_to when already a multiple
        else:
            Fp = F_
        out = (torch.zeros if Fp != F_ else torch.empty)(B, rows, Fp, dtype=torch.float32, device=x.device)
        lengths = None
        if seq_len is not None:
            lengths = seq_len.to(device=x.device, dtype=torch.int32).reshape(-1).contiguous()
            if lengths.numel() != B:
                raise ValueError("seq_len must hold one sample count per utterance")
        lo, hi = self._win_support
        call("fbank_forward_masked", x, _ll(x.stride(0)), B, N, lengths, self._window_full,
             self._twiddle, self.fb, self._fb_range, self.n_fft, lo, hi, self.hop_length,
             self.n_filt, float(self.preemph if self.preemph is not None else 0.0),
             int(bool(self.log)), out, _ll(rows * Fp), _ll(Fp), F_, self.frame_splicing,
             _ll(self.n_filt * Fp), self._NORMALIZE[self.normalize], int(n_signal))
        return out

    @classmethod
    def from_config(cls, cfg, log=False):
        """parts/features.py:349-357."""
        return cls(sample_rate=cfg['sample_rate'], window_size=cfg['window_size'],
                   window_stride=cfg['window_stride'], n_fft=cfg['n_fft'], nfilt=cfg['features'],
                   window=cfg['window'], normalize=cfg['normalize'],
                   max_duration=cfg.get('max_duration', 16.7), dither=cfg['dither'],
                   pad_to=cfg.get("pad_to", 0), frame_splicing=cfg.get("frame_splicing", 1), log=log)
