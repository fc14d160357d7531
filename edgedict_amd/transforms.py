"""Feature-transform builders with the reference's names (rnnt/transforms.py).

Only the log-mel path of the north-star is implemented on the GPU: ``FilterbankFeatures`` and the
frame stacking ``Downsample`` (pure data movement).  ``build_transform`` returns the same triple
``(transform_train, transform_test, input_size)`` as the reference (rnnt/transforms.py:165-203):
the train transform carries the SpecAugment masks (``SpecAugment``, one kernel on the resident
stacked batch), the test transform does not; both accept the reference's per-utterance call
``t(x)`` and the batched ``t(wave, wave_len)``.  Deltas and CMVN are outside the hot path
 and raise if requested.
"""
import torch

from .features import FilterbankFeatures, StackedLogFbank


class Downsample(torch.nn.Module):
    """Stack ``n_frame`` consecutive frames: [B, D, F] -> [B, D*n_frame, ceil(F/n_frame)]
    (rnnt/transforms.py:30-51).  View/pad/reshape only — no arithmetic."""

    def __init__(self, n_frame, pad_to_divisible=True):
        super().__init__()
        self.n_frame = n_frame
        self.pad_to_divisible = pad_to_divisible

    @torch.no_grad()
    def forward(self, feat):
        feat = feat.transpose(1, 2)
        B, L, D = feat.shape
        k = self.n_frame
        if self.pad_to_divisible:
            pad = (k - L % k) % k
            if pad:
                feat = torch.cat([feat, feat.new_zeros(B, pad, D)], dim=1)
        else:
            feat = feat[:, :L - L % k]
        return feat.reshape(B, -1, D * k).transpose(1, 2)


class _FusedFbankDownsample(torch.nn.Module):
    """What ``Sequential(FilterbankFeatures, Downsample)`` computes, in one kernel; output keeps
    the reference layout [B, D*n_frame, T0] so callers can still ``.transpose(1, 2)`` it
    (rnnt/stream.py:96, rnnt/dataset.py:103)."""

    def __init__(self, n_frame, pad_to_divisible, augment=None, **fb):
        super().__init__()
        self.inner = StackedLogFbank(n_frame=n_frame, pad_to_divisible=pad_to_divisible, **fb)
        self.augment = augment          # SpecAugment of the TRAIN transform (rnnt/transforms.py:196-200)

    def forward(self, x, lengths=None):
        """``forward(x)``: the reference's per-utterance call, returns [B, D*n_frame, T0].
        ``forward(wave [B,N], wave_len [B])``: the batched form for a trainer that collates raw
        audio (``wave_collate``), returns ``(xs [B, T0, D*n_frame], xlen [B])`` time-major, the
        lengths on the host when they came from the host."""
        xs, xlen = self.inner(x, lengths)
        if self.augment is not None:
            xs = self.augment(xs, xlen if lengths is not None else None)
        if lengths is None:
            return xs.transpose(1, 2)
        return xs, xlen


class SpecAugment(torch.nn.Module):
    """``TimeMasking(T_mask, T_num_mask)`` followed by ``FrequencyMasking(F_mask, F_num_mask)``
    (rnnt/transforms.py:54-146, zero fill) on a resident, already stacked feature batch
    ``xs [B, T0, F]`` — the reference runs them in DataLoader workers on ``[B, F, T0]``.

    The intervals are drawn with Python's ``random`` in exactly the reference's order (all time
    masks row by row, then all frequency masks row by row; ``start = randrange(dim)``,
    ``end = start + randrange(max_width)``), so ``random.seed(s)`` gives the reference's masks;
    one kernel applies them in place."""

    def __init__(self, T_mask=0, T_num_mask=0, F_mask=0, F_num_mask=0):
        super().__init__()
        self.T_mask, self.T_num_mask = T_mask, T_num_mask
        self.F_mask, self.F_num_mask = F_mask, F_num_mask

    def draw(self, B, T0, F, xlen=None):
        """Half-open [start, end) intervals, int32 CPU tensors [B, n, 2] (or None).  ``xlen`` (host
        integers, one per row): each row's time masks start inside ITS frames, as in the reference, which
        masks every utterance on its own length before the batch is padded (rnnt/dataset.py:98-104)."""
        import random
        t_iv = f_iv = None
        if self.T_mask > 0 and self.T_num_mask > 0:
            rows = []
            for b in range(B):
                Tb = T0 if xlen is None else max(1, min(T0, int(xlen[b])))
                for _ in range(self.T_num_mask):
                    start = random.randrange(0, Tb)
                    rows.append((start, start + random.randrange(0, self.T_mask)))
            t_iv = torch.tensor(rows, dtype=torch.int32).view(B, self.T_num_mask, 2)
        if self.F_mask > 0 and self.F_num_mask > 0:
            rows = []
            for _ in range(B):
                for _ in range(self.F_num_mask):
                    start = random.randrange(0, F)
                    rows.append((start, start + random.randrange(0, self.F_mask)))
            f_iv = torch.tensor(rows, dtype=torch.int32).view(B, self.F_num_mask, 2)
        return t_iv, f_iv

    @torch.no_grad()
    def forward(self, xs, xlen=None):
        from . import ops
        B, T0, F = xs.shape
        if xlen is not None and getattr(xlen, "is_cuda", False):
            xlen = None              # lengths on the device: no sync for them, whole-batch frame count
        t_iv, f_iv = self.draw(B, T0, F, None if xlen is None else [int(v) for v in xlen])
        if t_iv is None and f_iv is None:
            return xs
        dev = xs.device
        from ._staging import to_device
        return ops.spec_mask_(xs, None if t_iv is None else to_device(t_iv, dev),
                              None if f_iv is None else to_device(f_iv, dev))


def build_transform(feature_type, feature_size, n_fft=512, win_length=400, hop_length=200,
                    delta=False, cmvn=False, downsample=1, T_mask=0, T_num_mask=0, F_mask=0,
                    F_num_mask=0, pad_to_divisible=True, dither=1e-5):
    if feature_type != 'logfbank':
        raise NotImplementedError("only feature_type='logfbank' runs on the MI355X hot path "
                                  "(mfcc / melspec are torchaudio CPU transforms in the reference)")
    if delta:
        raise NotImplementedError("delta features are not part of the hot path (SURVEY.md 8f)")
    fb = dict(n_filt=feature_size, n_fft=n_fft, win_length=win_length, hop_length=hop_length,
              dither=dither)
    input_size = feature_size
    aug = None
    if (T_mask > 0 and T_num_mask > 0) or (F_mask > 0 and F_num_mask > 0):
        aug = SpecAugment(T_mask, T_num_mask, F_mask, F_num_mask)
    if downsample > 1:
        test = _FusedFbankDownsample(downsample, pad_to_divisible, **fb)
        train = _FusedFbankDownsample(downsample, pad_to_divisible, augment=aug, **fb) if aug else test
        input_size = input_size * downsample
    else:
        if aug is not None:
            raise NotImplementedError("SpecAugment masks are implemented on the stacked features "
                                      "(downsample > 1), as the shipped flagfiles use them")
        test = train = FilterbankFeatures(**fb)
    # the masks belong to the train transform only (rnnt/transforms.py:193-201)
    return train, test, input_size
