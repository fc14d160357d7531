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


def test_compiled_reference_loops_match_the_reference_checkout():
    """oracle/_ref/*.bin (oracle/ref_lift.py) is what tests/test_reference_loops_gpu.py executes on the GPU box, where
    /root/reference does not exist: here, where it does, the manifest must name the current reference files (a stale
    build would test yesterday's loops) and every piece must load into code that defines the names it promises."""
    import hashlib
    import json
    import os
    import pytest
    from oracle import ref_lift
    if not os.path.isdir(ref_lift.REF):
        pytest.skip("no reference checkout here")
    if not ref_lift.available():
        ref_lift.build(verbose=False)
    man = json.load(open(os.path.join(ref_lift.OUT, "manifest.json")))
    for name, (rel, cls, names) in ref_lift.PIECES.items():
        entry = man["pieces"][name]
        digest = hashlib.sha256(open(os.path.join(ref_lift.REF, rel), "rb").read()).hexdigest()
        assert entry["sha256_of_reference_file"] == digest, name
        ns = ref_lift.load(name, {"torch": __import__("torch")})
        assert all(n in ns for n in names), name
