"""``from rnnt.features import FilterbankFeatures`` -> fused HIP front-end (reference rnnt/features.py)."""
from edgedict_amd.features import FilterbankFeatures, StackedLogFbank  # noqa: F401
