"""GPU parity of the joint network's broadcast-add + tanh kernels (reference arithmetic:
rnnt/models.py:169-179, first Linear split as W1e*enc + W1d*dec) against a float64
torch restatement, on ragged shapes: label counts above the kernel's 72-position pass, joint sizes
that leave a partial 64-wide block, frame counts that leave partial slabs."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(2, 5, 3, 8), (3, 11, 21, 72), (2, 37, 80, 136), (4, 9, 65, 640), (1, 1, 1, 64)]


def _ref_bwd(dhid, hid):
    dp = dhid.double() * (1.0 - hid.double() ** 2)
    return dp.sum(2), dp.sum(1)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_joint_hidden_fwd_bwd(hip_lib, shape, dtype):
    from edgedict_amd import ops
    B, T, U1, J = shape
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + T * 100 + U1 + J)
    E1 = torch.randn(B, T, J, generator=g).to(dtype).cuda()
    D1 = torch.randn(B, U1, J, generator=g).to(dtype).cuda()
    hid = ops.joint_hidden_fwd(E1, D1)
    ref = torch.tanh(E1.double()[:, :, None] + D1.double()[:, None])
    tol = 1e-6 if dtype == torch.float32 else 8e-3
    assert (hid.double() - ref).abs().max().item() < tol
    dhid = torch.randn(B, T, U1, J, generator=g).to(dtype).cuda()
    dE1, dD1 = ops.joint_hidden_bwd(dhid, hid)
    rE, rD = _ref_bwd(dhid, hid)
    # fp32 accumulation of up to max(T, U1) products of O(1) terms (inputs are used as stored)
    assert (dE1.double() - rE).abs().max().item() < 1e-4 * max(1.0, rE.abs().max().item())
    assert (dD1.double() - rD).abs().max().item() < 1e-4 * max(1.0, rD.abs().max().item())
