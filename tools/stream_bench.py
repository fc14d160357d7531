"""Streaming decode throughput (BASELINE.json config 4): S concurrent streams on one GPU through
``BatchedStreamDecoder`` (the rnnt.stream loop of the reference, rnnt/stream.py:78-120, batched),
E6D2 model, bf16, reference-native chunk (win 1320 / hop 1200 samples = 75 ms -> 1 encoder frame
per chunk at step_n_frame = 2).  Reports stream-chunks/s and audio-seconds/s; the reference's README
quotes 5.8 audio-s/s for its CPU batch-1 decoder (README.md:125 region)."""
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
for dtype in (sys.argv[1:] or ("bf16", "fp32")):
    m.compute_dtype = dtype
    for step_n_frame in (2, 8):
        win, hop = chunk_geometry(flags, step_n_frame)
        for S in (1, 64, 256, 1024):
            dec = BatchedStreamDecoder(m, flags, S)
            chunk = 0.1 * torch.randn(S, win, device="cuda")
            n = 20 if S <= 256 else 8
            for _ in range(3):
                dec.decode(chunk)
            torch.cuda.synchronize()
            t = time.time()
            for _ in range(n):
                toks = dec.decode(chunk)
            torch.cuda.synchronize()
            dt = (time.time() - t) / n
            print("%s chunk %4d ms (%d encoder frames)  S=%4d: %7.2f ms per chunk step -> %8.0f stream-chunks/s, %8.0f audio-s/s (%.0fx real time per stream)"
                  % (dtype, hop / 16, toks.shape[1], S, dt * 1e3, S / dt, S * hop / 16000 / dt, hop / 16000 / dt))
