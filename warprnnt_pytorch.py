"""Import shim: ``from warprnnt_pytorch import RNNTLoss`` (reference rnnt/models.py:8-11,
cli/lightning.py:12) resolves to the gfx950 loss kernels."""
from edgedict_amd.loss import RNNTLoss  # noqa: F401
