"""``from rnnt.stream import PytorchStreamDecoder`` -> MI355X engine (reference rnnt/stream.py)."""
from edgedict_amd.stream import (StreamTransducerDecoder, PytorchStreamDecoder,  # noqa: F401
                                 BatchedStreamDecoder, chunk_geometry)
