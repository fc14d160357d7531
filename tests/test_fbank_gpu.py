"""GPU parity of the fused log-mel kernel against the CPU oracle (dither off)."""
import os

import numpy as np
import pytest
import torch

from oracle import features_ref as Fr

pytestmark = pytest.mark.gpu


def _audio(B, N, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (0.1 * torch.randn(B, N, generator=g)).clamp_(-1, 1)


@pytest.mark.parametrize("N,hop,win", [(80000, 160, 320), (24000, 200, 320), (12345, 320, 400),
                                       (1320, 200, 320), (513, 160, 320)])
def test_filterbank_module_matches_oracle(hip_lib, N, hop, win):
    from edgedict_amd.features import FilterbankFeatures
    x = _audio(3, N, N)
    ref = Fr.log_fbank(x, win_length=win, hop_length=hop)
    fb = FilterbankFeatures(win_length=win, hop_length=hop, n_filt=80, dither=0).cuda()
    out = fb(x.cuda()).cpu()
    assert out.shape == ref.shape
    # fp32 FFT vs MKL FFT: compare log-energies; random-noise audio has no near-zero bins
    assert (out - ref).abs().max().item() < 2e-4


def test_stacked_layout_ragged_lengths_and_mask(hip_lib):
    from edgedict_amd.features import StackedLogFbank
    x = _audio(4, 80000, 7)
    lengths = torch.tensor([80000, 64000, 48123, 1600], dtype=torch.int32)
    mod = StackedLogFbank(n_frame=3, win_length=320, hop_length=160, n_filt=80, dither=0).cuda()
    xs, xlen = mod(x.cuda(), lengths.cuda())
    assert xs.shape == (4, 167, 240)
    for b in range(4):
        n = int(lengths[b])
        ref = Fr.stacked_features(x[b:b + 1, :n], hop_length=160)[0]     # per-utterance, as the dataset does
        assert int(xlen[b]) == ref.shape[0]
        got = xs[b].cpu()
        assert (got[:ref.shape[0]] - ref).abs().max().item() < 2e-4
        assert got[ref.shape[0]:].abs().max().item() == 0 if ref.shape[0] < 167 else True


def test_streaming_chunk_truncates_instead_of_padding(hip_lib):
    from edgedict_amd.features import StackedLogFbank
    # reference streaming chunk at E6D2: win_size 1320 -> 7 STFT frames -> 2 stacked frames
    x = _audio(2, 1320, 3)
    mod = StackedLogFbank(n_frame=3, pad_to_divisible=False, win_length=320, hop_length=200,
                          n_filt=80, dither=0).cuda()
    xs, xlen = mod(x.cuda())
    ref = Fr.stacked_features(x, 3, False, hop_length=200)
    assert xs.shape == ref.shape == (2, 2, 240)
    assert (xs.cpu() - ref).abs().max().item() < 2e-4


def test_dither_statistics_and_in_place_side_effect(hip_lib):
    from edgedict_amd.features import FilterbankFeatures
    x = torch.zeros(2, 160000).cuda()
    fb = FilterbankFeatures(win_length=320, hop_length=160, n_filt=80, dither=1e-5).cuda()
    fb(x)
    # the caller's tensor was modified in place, like rnnt/features.py:111-112
    assert abs(x.std().item() - 1e-5) < 2e-7 and abs(x.mean().item()) < 1e-7
    k = ((x / 1e-5) ** 4).mean().item()
    assert abs(k - 3.0) < 0.1          # Gaussian kurtosis
    y = torch.zeros(2, 160000).cuda()
    fb(y)
    assert not torch.equal(x, y)       # fresh noise per call


GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "features.npz"))


def _wave(seed, B, N):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return 0.1 * torch.randn(B, N, generator=g)


def test_filterbank_and_stacking_match_reference_executed_goldens(hip_lib):
    """tests/golden/features.npz: outputs of the reference's own FilterbankFeatures.forward
    (rnnt/features.py:106-152) and Downsample.forward (rnnt/transforms.py:38-51)."""
    from edgedict_amd.features import FilterbankFeatures, StackedLogFbank
    from oracle.make_golden_features import RNNT_CASES
    for i, (seed, B, N, win, hop, nf, stride) in enumerate(RNNT_CASES):
        x = _wave(seed, B, N)
        fb = FilterbankFeatures(win_length=win, hop_length=hop, n_fft=512, n_filt=nf, dither=0).cuda()
        out = fb(x.cuda()).cpu().numpy()
        ref = GOLD["rnnt%d_feat" % i]
        assert out[:, :, ::stride].shape == ref.shape
        assert np.abs(out[:, :, ::stride] - ref).max() < 2e-4, i
        if stride == 1:
            for pad, tag in ((True, "pad"), (False, "trunc")):
                mod = StackedLogFbank(n_frame=3, pad_to_divisible=pad, win_length=win, hop_length=hop,
                                      n_fft=512, n_filt=nf, dither=0).cuda()
                xs, xlen = mod(x.cuda())
                z = GOLD["rnnt%d_stack_%s" % (i, tag)]            # reference layout [B, 3*n_filt, T0]
                assert xs.shape == (B, z.shape[2], z.shape[1])
                assert np.abs(xs.cpu().numpy().transpose(0, 2, 1) - z).max() < 2e-4, (i, tag)
                assert int(xlen[0]) == z.shape[2]


def test_parts_twin_matches_reference_executed_goldens(hip_lib):
    """parts/features.py:228-357 (seconds-based ctor, forward(x, seq_len), windows, normalize_batch,
    frame splicing, short-input zero-pad, pad_to) against the reference's own class."""
    from parts.features import FilterbankFeatures
    from oracle.make_golden_features import PARTS_CASES
    for i, (kw, seed, B, N, seq) in enumerate(PARTS_CASES):
        x = _wave(seed, B, N)
        m = FilterbankFeatures(**kw).cuda()
        out = m(x.cuda(), torch.tensor(seq, dtype=torch.int32).cuda()).cpu().numpy()
        ref = GOLD["parts%d_feat" % i]
        assert out.shape == ref.shape, (i, out.shape, ref.shape)
        # normalised features are O(1); un-normalised log-energies O(10): same absolute bound as above
        tol = 2e-4 if kw["normalize"] == "none" else 5e-4
        assert np.abs(out - ref).max() < tol, (i, np.abs(out - ref).max())


def test_parts_twin_rejects_what_the_reference_cannot_pad(hip_lib):
    from parts.features import FilterbankFeatures
    m = FilterbankFeatures(sample_rate=16000, window_size=0.02, window_stride=0.01, dither=0.0).cuda()
    with pytest.raises(RuntimeError):      # 320 < N = 400 < n_fft = 512 (parts/features.py:289-294)
        m(torch.zeros(1, 400).cuda(), torch.tensor([400]).cuda())
