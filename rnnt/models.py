"""``from rnnt.models import Transducer`` -> MI355X engine (replaces reference rnnt/models.py)."""
from edgedict_amd.models import (Transducer, Encoder, Decoder, Joint, ResLayerNormLSTM,  # noqa: F401
                                 TimeReduction, convert_lightning2normal,
                                 FrontEnd, ResLayerNormGRU, CTCEncoder)
from edgedict_amd.loss import RNNTLoss  # noqa: F401
