"""Special token ids of the hot-path contract (reference: rnnt/tokenizer.py:7-10)."""
NUL = 0   # the RNN-T blank
PAD = 1   # label padding; embedding padding_idx
BOS = 2   # prediction-network start symbol
UNK = 3
