"""``from rnnt.transforms import build_transform`` -> MI355X engine (reference rnnt/transforms.py)."""
from edgedict_amd.transforms import Downsample, build_transform  # noqa: F401
from rnnt import _reference_fallback  # noqa: E402

# names the engine does not provide (corpus readers, audio-file transforms, wav2vec pieces ...) fall
# through to the reference checkout when one is on sys.path
__getattr__ = _reference_fallback("transforms", __file__)
