"""CPU: closed-form pins for the log-mel oracle (the reference's rnnt.features cannot run here)."""
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
