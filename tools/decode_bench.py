"""Decode throughput on the E6D2 shape (run on the GPU box): batched greedy search and batched beam
search (W = 10) over 64 x 15 s utterances, bf16.  Random weights give near-uniform output
distributions, for which Graves' search pops ~V/2 hypotheses per frame; a trained model emits blank
on most frames, so the blank logit is biased up here and every frame costs the minimum of W
expansions - the regime the committed trained tiny model shows (tests/golden/beam_tiny.npz: exactly
W pops per frame)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from edgedict_amd import decode  # noqa: E402
from edgedict_amd.flags import make_flags, model_kwargs  # noqa: E402
from edgedict_amd.models import Transducer  # noqa: E402

flags = make_flags("E6D2")
torch.manual_seed(0)
m = Transducer(**model_kwargs(flags, vocab_size=2048)).cuda().eval()
m.compute_dtype = "bf16"
with torch.no_grad():
    m.joint.joint[2].bias[0] += 12.0
B, T0 = 64, 401
xs = torch.randn(B, T0, flags.feature_size * flags.downsample, device="cuda")
xlen = torch.full((B,), T0, dtype=torch.int32)


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n


with torch.no_grad():
    tg = timed(lambda: m.greedy_decode(xs, xlen))
    print("greedy: %.1f ms per batch of %d  -> %.0f utt/s (%.0f x real time)" % (tg * 1e3, B, B / tg, B * 15 / tg))
    for W in (4, 10):
        tb = timed(lambda: m.beam_search(xs, xlen, W=W), n=2)
        n = decode.beam_search_batch.last_expansions
        print("beam W=%d: %.1f ms per batch of %d -> %.0f utt/s (%.0f x real time), %d expansions, %.2f M expansions/s, %.1f us per lockstep iteration"
              % (W, tb * 1e3, B, B / tb, B * 15 / tb, n, n / tb / 1e6, tb * 1e6 / (n / B)))
