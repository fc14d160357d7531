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
