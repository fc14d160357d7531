"""Special-token ids of the hot-path contract (reference rnnt/tokenizer.py:7-10)."""
from edgedict_amd.tokenizer import NUL, PAD, BOS, UNK  # noqa: F401
