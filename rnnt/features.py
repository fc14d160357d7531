"""``from rnnt.features import FilterbankFeatures`` -> fused HIP front-end (reference rnnt/features.py)."""
from edgedict_amd.features import FilterbankFeatures, StackedLogFbank  # noqa: F401
from rnnt import _reference_fallback  # noqa: E402

# names the engine does not provide (corpus readers, audio-file transforms, wav2vec pieces ...) fall
# through to the reference checkout when one is on sys.path
__getattr__ = _reference_fallback("features", __file__)
