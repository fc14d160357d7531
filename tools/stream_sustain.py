"""Does the streaming chunk step hold its rate over long runs?  ms per chunk step for n = 20 .. 800 consecutive chunk steps
(S streams, E6D2, bf16, 75 ms chunks), with and without a device synchronisation every 20 steps.
usage: python tools/stream_sustain.py [S]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from edgedict_amd.flags import make_flags, model_kwargs  # noqa: E402
from edgedict_amd.models import Transducer  # noqa: E402
from edgedict_amd.stream import BatchedStreamDecoder, chunk_geometry  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
flags = make_flags("E6D2")
torch.manual_seed(0)
m = Transducer(**model_kwargs(flags, vocab_size=2048)).cuda().eval()
m.compute_dtype = "bf16"
win, hop = chunk_geometry(flags, 2)
dec = BatchedStreamDecoder(m, flags, S)
chunk = 0.1 * torch.randn(S, win, device="cuda")
for _ in range(5):
    dec.decode(chunk)
torch.cuda.synchronize()
for n in (20, 50, 100, 200, 400, 800, 20):
    t = time.time()
    for _ in range(n):
        dec.decode(chunk)
    th = time.time() - t
    torch.cuda.synchronize()
    ta = time.time() - t
    print("n = %4d: host %.3f ms per step, device done %.3f ms per step" % (n, th / n * 1e3, ta / n * 1e3))
t = time.time()
for i in range(400):
    dec.decode(chunk)
    if i % 20 == 19:
        torch.cuda.synchronize()
print("n =  400 with a synchronize every 20 steps: %.3f ms per step" % ((time.time() - t) / 400 * 1e3))
