"""CPU: pins for the log-mel oracle - tests/golden/features.npz holds outputs of the REFERENCE's own
FilterbankFeatures (rnnt/features.py and the parts/features.py twin) and Downsample classes,
executed by oracle/make_golden_features.py; plus closed-form properties."""
import os
import math

import numpy as np
import torch

from oracle import features_ref as Fr


def test_frame_counts_match_survey_a1():
    assert Fr.log_fbank(torch.zeros(1, 80000), hop_length=160).shape == (1, 80, 501)
    assert Fr.log_fbank(torch.zeros(1, 240000), hop_length=200).shape == (1, 80, 1201)
    assert Fr.stacked_features(torch.zeros(1, 80000), hop_length=160).shape == (1, 167, 240)
    assert Fr.stacked_features(torch.zeros(1, 240000), hop_length=200).shape == (1, 401, 240)


def test_last_frame_masked_when_hop_divides_n_and_silence_is_log_floor():
    f = Fr.log_fbank(torch.zeros(1, 1600), hop_length=160)
    assert f.shape[-1] == 11
    assert torch.all(f[..., -1] == 0)
    np.testing.assert_allclose(f[..., :-1].numpy(), math.log(1e-20), rtol=1e-6)


def test_slaney_mel_scale_known_points():
    assert abs(Fr.hz_to_mel(1000.0) - 15.0) < 1e-12
    assert abs(Fr.hz_to_mel(500.0) - 7.5) < 1e-12
    assert abs(Fr.mel_to_hz(Fr.hz_to_mel(6400.0)) - 6400.0) < 1e-9
    assert abs(Fr.hz_to_mel(6400.0) - 42.0) < 1e-9      # 15 + 27 by construction of the scale


def test_mel_filters_are_area_normalised_triangles():
    fb = Fr.mel_filters(16000, 512, 80).astype(np.float64)
    assert fb.shape == (80, 257) and fb.min() >= 0
    # each triangle integrates to ~1 over frequency (bin width 31.25 Hz) once it spans enough bins
    area = fb.sum(1) * 31.25
    assert np.all(np.abs(area - 1.0) < 0.12)       # coarse bins at the low end
    assert np.all(np.abs(area[60:] - 1.0) < 0.01)  # wide triangles: discretisation error vanishes
    peak = fb.argmax(1)
    assert np.all(np.diff(peak) >= 0)                    # centre frequencies increase


def test_product_mel_table_equals_oracle_table():
    from edgedict_amd.features import mel_filterbank
    a = mel_filterbank(16000, 512, 80)
    np.testing.assert_allclose(a, Fr.mel_filters(16000, 512, 80), atol=1e-7)


def test_downsample_layout_and_streaming_truncation():
    feat = torch.arange(2 * 4 * 7, dtype=torch.float32).reshape(2, 4, 7)   # [B, mel, F]
    z = Fr.downsample(feat, 3, True)
    assert z.shape == (2, 12, 3)
    # z[b, k*4+m, tau] = feat[b, m, 3*tau+k]; padded frames are zero
    assert z[1, 1 * 4 + 2, 1] == feat[1, 2, 4]
    assert torch.all(z[:, 4:, 2] == 0)
    assert Fr.downsample(feat, 3, False).shape == (2, 12, 2)


GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "features.npz"))


def _wave(seed, B, N):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return 0.1 * torch.randn(B, N, generator=g)


def test_mel_table_matches_third_party_librosa_equivalent():
    """golden mel tables come from HuggingFace transformers' mel_filter_bank(norm='slaney',
    mel_scale='slaney'), an independent implementation of librosa.filters.mel."""
    from edgedict_amd.features import mel_filterbank
    for key, (sr, nfft, n) in {"mel_80": (16000, 512, 80), "mel_64": (16000, 512, 64),
                               "mel_8k_64": (8000, 256, 64)}.items():
        np.testing.assert_allclose(Fr.mel_filters(sr, nfft, n), GOLD[key], atol=2e-9)
        np.testing.assert_allclose(mel_filterbank(sr, nfft, n), GOLD[key], atol=1e-7)


def test_oracle_reproduces_reference_filterbank_and_downsample_outputs():
    from oracle.make_golden_features import RNNT_CASES
    for i, (seed, B, N, win, hop, nf, stride) in enumerate(RNNT_CASES):
        assert list(GOLD["rnnt%d_cfg" % i]) == [seed, B, N, win, hop, nf, stride]
        x = _wave(seed, B, N)
        mine = Fr.log_fbank(x, win_length=win, hop_length=hop, n_fft=512, n_filt=nf)
        np.testing.assert_allclose(mine.numpy()[:, :, ::stride], GOLD["rnnt%d_feat" % i], atol=2e-5)
        if stride == 1:
            for pad, tag in ((True, "pad"), (False, "trunc")):
                z = Fr.downsample(mine, 3, pad)
                np.testing.assert_allclose(z.numpy(), GOLD["rnnt%d_stack_%s" % (i, tag)], atol=2e-5)


def test_oracle_reproduces_reference_parts_twin_outputs():
    from oracle.make_golden_features import PARTS_CASES
    for i, (kw, seed, B, N, seq) in enumerate(PARTS_CASES):
        x = _wave(seed, B, N)
        mine = Fr.parts_log_fbank(x, torch.tensor(seq, dtype=torch.int32), **kw)
        ref = GOLD["parts%d_feat" % i]
        assert mine.shape == ref.shape
        np.testing.assert_allclose(mine.numpy(), ref, atol=5e-5)
