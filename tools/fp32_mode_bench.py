"""The token-exact fp32 mode at the E6D2 bench geometry: greedy decode of 64 x 15 s, the encoder alone, and 256 streams
(EDGEDICT_LSTM_F32_LPW=0/1 switches the launch-persistent fp32 recurrence).   python tools/fp32_mode_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from edgedict_amd import side  # noqa: E402
from edgedict_amd.features import StackedLogFbank  # noqa: E402
from edgedict_amd.flags import make_flags, model_kwargs  # noqa: E402
from edgedict_amd.models import Transducer  # noqa: E402

dev = torch.device("cuda", 0)
side.stream(dev)
flags = make_flags("E6D2", gradclip=None, dither=1e-5)
torch.manual_seed(0)
m = Transducer(**model_kwargs(flags, vocab_size=flags.bpe_size)).to(dev).eval()
m.compute_dtype = "fp32"
wave, wave_len, ys, ylen = bench.synth_batch(flags, 64, 15.0, 64, 1000, dev)
fb = StackedLogFbank(n_frame=flags.downsample, pad_to_divisible=True, out_dtype=torch.float32, sample_rate=16000,
                     win_length=flags.win_length, hop_length=flags.hop_length, n_fft=flags.n_fft, n_filt=flags.feature_size,
                     dither=0.0).to(dev)
with torch.no_grad():
    xs, xlen = fb(wave, wave_len)
    for name, fn in (("encoder fp32", lambda: m.encoder(xs)), ("greedy fp32", lambda: m.greedy_decode(xs, xlen))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print("%-14s %7.2f ms per batch of 64  -> %6.0f utt/s" % (name, 1e3 * dt, 64 / dt), flush=True)
r = bench.stream_256(flags, dev, dtype="fp32", n_chunks=20)
print("stream_256 fp32: %.3f ms per chunk step" % r["ms_per_chunk_step"])
