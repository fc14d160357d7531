"""GPU: SpecAugment masking kernel vs the reference-pinned golden masks, and dropout
(nn.Dropout / nn.LSTM(dropout=p) training semantics, rnnt/models.py:47-53,145-147)."""
import os
import random

import pytest
import torch

from oracle import transforms_ref as Tr

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "specaug.npz")


def cases():
    return Tr.golden_cases(GOLD)


def test_spec_augment_kernel_reproduces_reference_masks(hip_lib):
    from edgedict_amd.transforms import SpecAugment
    for seed, x, mask, cfg in cases():
        xs = x.transpose(1, 2).contiguous().cuda()      # engine layout [B, T0, F]
        random.seed(seed)
        out = SpecAugment(*cfg)(xs)
        assert out.data_ptr() == xs.data_ptr()           # in place on the resident batch
        ref = x.masked_fill(mask, 0).transpose(1, 2)
        assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dropout_statistics_scaling_and_backward(hip_lib, dtype):
    from edgedict_amd.models import _DropoutFn
    p = 0.1
    x = torch.ones(64, 65, 512, dtype=dtype, device="cuda", requires_grad=True)
    y = _DropoutFn.apply(x, p, 1234)
    kept = (y != 0)
    frac = kept.float().mean().item()
    assert abs(frac - (1 - p)) < 3e-3                      # 2.1 M Bernoulli draws
    vals = y[kept].float()
    assert (vals - 1.0 / (1 - p)).abs().max().item() < 1e-2   # bf16 rounding of 1.111
    g = torch.full_like(y, 2.0)
    y.backward(g)
    assert torch.equal(x.grad != 0, kept)                  # same mask in backward
    assert (x.grad[kept].float() - 2.0 / (1 - p)).abs().max().item() < 2e-2
    y2 = _DropoutFn.apply(x.detach(), p, 1234)
    y3 = _DropoutFn.apply(x.detach(), p, 1235)
    assert torch.equal(y2, y.detach()) and not torch.equal(y3, y.detach())
    # no structure along rows/columns: every row keeps about 90 %
    rows = kept.float().mean(dim=(0, 2))
    assert (rows - (1 - p)).abs().max().item() < 0.02


def test_prediction_network_dropout_trains_and_is_identity_in_eval(hip_lib):
    """E6D2_LARGE_Batch uses dec_dropout=0.1 (flagfiles/E6D2_LARGE_Batch.txt)."""
    from edgedict_amd.models import Decoder
    torch.manual_seed(0)
    dec = Decoder(vocab_embed_size=16, vocab_size=50, hidden_size=32, num_layers=2, dropout=0.1,
                  proj_size=24).cuda()
    ys = torch.randint(4, 50, (3, 6), dtype=torch.int32).cuda()
    dec.eval()
    with torch.no_grad():
        a, _ = dec(ys)
        b, _ = dec(ys)
    assert torch.equal(a, b)
    dec.train()
    c, _ = dec(ys)
    d, _ = dec(ys)
    assert not torch.equal(c, d)                           # fresh mask per call
    assert (c - a).abs().max().item() > 0
    c.square().sum().backward()
    for n, q in dec.named_parameters():
        assert q.grad is not None and torch.isfinite(q.grad).all(), n


def test_build_transform_triple_train_masks_test_does_not(hip_lib):
    """rnnt/transforms.py:165-203: (transform_train, transform_test, input_size); the masks belong
    to the train transform only.  Both call forms: per-utterance ``t(x)`` -> [B, D*n, T0] (reference
    layout) and batched ``t(wave, wave_len)`` -> (xs [B, T0, D*n], xlen)."""
    from edgedict_amd.transforms import build_transform
    kw = dict(feature_type="logfbank", feature_size=80, n_fft=512, win_length=320, hop_length=200,
              downsample=3)
    train, test, input_size = build_transform(T_mask=50, T_num_mask=2, F_mask=5, F_num_mask=1, **kw)
    assert input_size == 240 and train is not test
    plain_train, plain_test, _ = build_transform(**kw)
    assert plain_train is plain_test                         # no masks requested
    g = torch.Generator().manual_seed(0)
    wave = (0.1 * torch.randn(3, 16000, generator=g)).cuda()
    wl = torch.tensor([16000, 12000, 9000], dtype=torch.int32)
    train, test = train.cuda(), test.cuda()                  # nn.Modules: buffers follow .to(device)
    for t in (train, test):
        t.inner.fbank.dither = 0.0
    ref = test(wave.clone())
    assert ref.shape[0] == 3 and ref.shape[1] == 240
    xs_t, xlen = test(wave.clone(), wl)
    assert xs_t.shape == (3, ref.shape[2], 240) and not xlen.is_cuda
    assert torch.equal(xs_t[0].t(), ref[0])                  # same features, time-major
    random.seed(5)
    xs_a, xlen_a = train(wave.clone(), wl)
    assert torch.equal(xlen_a, xlen)
    zeroed = (xs_a == 0) & (xs_t != 0)
    assert zeroed.any()                                      # something was masked ...
    assert torch.equal(xs_a[~zeroed], xs_t[~zeroed])         # ... and nothing else changed
