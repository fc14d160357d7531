"""``from rnnt.transforms import build_transform`` -> MI355X engine (reference rnnt/transforms.py)."""
from edgedict_amd.transforms import Downsample, build_transform  # noqa: F401
