"""CPU: the streaming-loop oracle (oracle/stream_ref.py) reproduces tests/golden/stream.npz - texts
returned by the REFERENCE's own PytorchStreamDecoder.reset/decode (rnnt/stream.py:78-120), executed
by oracle/make_golden_stream.py on the reference Transducer's sub-modules."""
import os

import numpy as np
import pytest
import torch

from oracle.make_golden_stream import CASES, HOP, WIN, StubVocab, ids_of, state_dict
from oracle.stream_ref import StreamOracle

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stream.npz"))


@pytest.mark.parametrize("name", ["small", "small_multi", "E6D2"])
def test_stream_oracle_reproduces_reference_texts(name):
    from edgedict_amd.flags import make_flags
    cfg, wseed, xseed, S, n_chunks, resets, bias = CASES[name]
    assert list(GOLD[name + "_cfg"]) == [wseed, xseed, S, n_chunks]
    flags = make_flags("E6D2")
    sd = state_dict(cfg, wseed, bias)
    g = torch.Generator(device="cpu").manual_seed(xseed)
    wave = 0.1 * torch.randn(S, WIN + n_chunks * HOP, generator=g)
    oracles = [StreamOracle(sd, flags) for _ in range(S)]
    texts = GOLD[name + "_texts"]
    vocab = StubVocab()
    for c in range(n_chunks):
        for s in resets.get(c, []):
            oracles[s].reset()
        for s in range(S):
            ids = [t for t in oracles[s].decode(wave[s:s + 1, c * HOP:c * HOP + WIN].clone()) if t != 0]
            assert ids == ids_of(str(texts[s, c])), (c, s)
            assert "".join(vocab.id_to_token(t).replace("</w>", " ") for t in ids) == str(texts[s, c])
