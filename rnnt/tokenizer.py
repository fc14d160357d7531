"""Special-token ids of the hot-path contract (reference rnnt/tokenizer.py:7-10) from the engine;
the tokenizer CLASSES (``HuggingFaceTokenizer``, ``CharTokenizer``: host-side text processing,
imported by cli/train.py:19 and rnnt/stream.py:12) come from the reference's own
``rnnt/tokenizer.py`` when a reference checkout is on ``sys.path``."""
from edgedict_amd.tokenizer import NUL, PAD, BOS, UNK  # noqa: F401
from rnnt import _reference_fallback

__getattr__ = _reference_fallback("tokenizer", __file__)
