"""ms per chunk step of BatchedStreamDecoder at S streams (E6D2, bf16, 75 ms chunks), 200 consecutive steps, 3 repeats.
usage: python tools/stream_step_time.py [S ...]      (environment switches, e.g. EDGEDICT_ENC_TILE_NS, apply)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from edgedict_amd.flags import make_flags, model_kwargs  # noqa: E402
from edgedict_amd.models import Transducer  # noqa: E402
from edgedict_amd.stream import BatchedStreamDecoder, chunk_geometry  # noqa: E402

flags = make_flags("E6D2")
torch.manual_seed(0)
m = Transducer(**model_kwargs(flags, vocab_size=2048)).cuda().eval()
m.compute_dtype = "bf16"
win, hop = chunk_geometry(flags, 2)
for S in [int(a) for a in sys.argv[1:]] or [256]:
    dec = BatchedStreamDecoder(m, flags, S)
    chunk = 0.1 * torch.randn(S, win, device="cuda")
    for _ in range(10):
        dec.decode(chunk)
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        t = time.time()
        for _ in range(200):
            dec.decode(chunk)
        torch.cuda.synchronize()
        best.append((time.time() - t) / 200 * 1e3)
    print("S = %4d: %.4f ms per chunk step (runs: %s)" % (S, min(best), ", ".join("%.4f" % b for b in best)))
