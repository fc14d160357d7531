"""Feature-transform builders with the reference's names (rnnt/transforms.py).

Only the log-mel path of the north-star is implemented on the GPU: ``FilterbankFeatures`` and the
frame stacking ``Downsample`` (pure data movement).  ``build_transform`` returns the same triple
``(transform_train, transform_test, input_size)`` as the reference (rnnt/transforms.py:165-203);
SpecAugment masking, deltas and CMVN are train-time augmentation outside the hot path
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

    def __init__(self, n_frame, pad_to_divisible, **fb):
        super().__init__()
        self.inner = StackedLogFbank(n_frame=n_frame, pad_to_divisible=pad_to_divisible, **fb)

    def forward(self, x):
        xs, _ = self.inner(x)
        return xs.transpose(1, 2)


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
    if downsample > 1:
        test = _FusedFbankDownsample(downsample, pad_to_divisible, **fb)
        input_size = input_size * downsample
    else:
        test = FilterbankFeatures(**fb)
    # SpecAugment masks (T_mask/F_mask) belong to the train transform only; they are applied by
    # the input pipeline, not by the engine -> train == test here.
    return test, test, input_size
