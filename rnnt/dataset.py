"""``from rnnt.dataset import seq_collate`` -> MI355X engine (reference rnnt/dataset.py:202-240).
Only the batch-assembly functions are provided: the corpus readers (LibriSpeech / CommonVoice /
TEDLIUM / YoutubeCaption index builders) are storage-side and out of scope."""
from edgedict_amd.collate import (  # noqa: F401
    PAD, end_pad_concat, seq_collate, wave_collate, zero_pad_concat)
from rnnt import _reference_fallback  # noqa: E402

# names the engine does not provide (corpus readers, audio-file transforms, wav2vec pieces ...) fall
# through to the reference checkout when one is on sys.path
__getattr__ = _reference_fallback("dataset", __file__)
