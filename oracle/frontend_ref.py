"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement, in functional PyTorch, of the reference's convolutional front-end:
    CausalConv1d       rnnt/models.py:314-318   Conv1d(padding=(k-1)*dilation) ...
    DilatedConvBlock   rnnt/models.py:323-339   GELU -> GroupNorm(1, C_in) -> conv -> drop last padding frames
    FrontEnd           rnnt/models.py:341-365   conv1 (+drop) -> blocks -> transpose to B x T x C -> LayerNorm(C)
taking a plain ``state_dict`` with the reference's key names (conv1.*, encode.{i}.conv.*,
encode.{i}.gn.*, layer_norm.*).

PARITY STATUS: **pinned** - ``oracle/make_golden_frontend.py`` runs the reference's own
``rnnt.models.FrontEnd`` on seeded weights / waveforms, checks this restatement against it and stores
the reference's outputs in ``tests/golden/frontend.npz``.
"""
import torch
import torch.nn.functional as F


def n_blocks(sd):
    n = 0
    while "encode.%d.conv.weight" % n in sd:
        n += 1
    return n


def causal_conv(x, w, b, stride):
    """x [B,C,T]; nn.Conv1d(padding=k-1, stride) then drop the last k-1 frames."""
    pad = w.shape[2] - 1
    y = F.conv1d(x, w, b, stride=stride, padding=pad)
    return y[:, :, :-pad] if pad > 0 else y


def group_norm_1(x, gamma, beta, eps=1e-5):
    """nn.GroupNorm(1, C): statistics over (C, T) of each sample, biased variance, affine per channel."""
    mean = x.mean(dim=(1, 2), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(1, 2), keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * gamma[None, :, None] + beta[None, :, None]


def frontend_forward(sd, x, strides):
    """x [B, N] waveform; ``strides`` = stride of conv1 and of every block (not in the state dict).
    Returns [B, T, C_last]."""
    if x.dim() < 3:
        x = x.unsqueeze(1)
    x = causal_conv(x, sd["conv1.weight"], sd.get("conv1.bias"), strides[0])
    for i in range(n_blocks(sd)):
        p = "encode.%d." % i
        x = F.gelu(x)
        x = group_norm_1(x, sd[p + "gn.weight"], sd[p + "gn.bias"])
        x = causal_conv(x, sd[p + "conv.weight"], sd.get(p + "conv.bias"), strides[i + 1])
    x = x.transpose(1, 2)
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + 1e-5) * sd["layer_norm.weight"] + sd["layer_norm.bias"]
