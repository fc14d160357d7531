"""``from rnnt.models import Transducer`` -> MI355X engine (replaces reference rnnt/models.py)."""
from edgedict_amd.models import (Transducer, Encoder, Decoder, Joint, ResLayerNormLSTM,  # noqa: F401
                                 TimeReduction, convert_lightning2normal,
                                 FrontEnd, ResLayerNormGRU, CTCEncoder)
from edgedict_amd.loss import RNNTLoss  # noqa: F401
from rnnt import _reference_fallback  # noqa: E402

# names the engine does not provide (corpus readers, audio-file transforms, wav2vec pieces ...) fall
# through to the reference checkout when one is on sys.path
__getattr__ = _reference_fallback("models", __file__)
