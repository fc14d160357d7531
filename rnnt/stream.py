"""``from rnnt.stream import PytorchStreamDecoder`` -> MI355X engine (reference rnnt/stream.py)."""
from edgedict_amd.stream import (StreamTransducerDecoder, PytorchStreamDecoder,  # noqa: F401
                                 BatchedStreamDecoder, chunk_geometry)
from rnnt import _reference_fallback  # noqa: E402

# names the engine does not provide (corpus readers, audio-file transforms, wav2vec pieces ...) fall
# through to the reference checkout when one is on sys.path
__getattr__ = _reference_fallback("stream", __file__)
