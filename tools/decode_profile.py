"""One decode workload under rocprofv3 (run on the GPU box):  python tools/decode_profile.py MODE
MODE = greedy_fp32 | greedy_bf16 | stream256 | beam10.  E6D2 model, 64 x 15 s utterances (stream: 256 streams,
75 ms chunks); two warm-up passes, then REPS timed ones - `tools/gpu_decode_profile.sh` wraps each mode in
`rocprofv3 --kernel-trace --stats` and summarises the kernels with profiles/summarize.py."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from edgedict_amd.flags import make_flags, model_kwargs  # noqa: E402
from edgedict_amd.models import Transducer  # noqa: E402
from edgedict_amd.stream import BatchedStreamDecoder, chunk_geometry  # noqa: E402

mode = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
flags = make_flags("E6D2")
torch.manual_seed(0)
m = Transducer(**model_kwargs(flags, vocab_size=2048)).cuda().eval()
with torch.no_grad():
    m.joint.joint[2].bias[0] += 12.0
B, T0 = 64, 401
xs = torch.randn(B, T0, flags.feature_size * flags.downsample, device="cuda")
xlen = torch.full((B,), T0, dtype=torch.int32)
if mode == "greedy_fp32":
    m.compute_dtype = "fp32"
    fn = lambda: m.greedy_decode(xs, xlen)
elif mode == "greedy_bf16":
    m.compute_dtype = "bf16"
    fn = lambda: m.greedy_decode(xs, xlen)
elif mode == "beam10":
    m.compute_dtype = "bf16"
    fn = lambda: m.beam_search(xs, xlen, W=10)
elif mode == "stream256":
    m.compute_dtype = "bf16"
    win, hop = chunk_geometry(flags, 2)
    dec = BatchedStreamDecoder(m, flags, 256)
    chunk = 0.1 * torch.randn(256, win, device="cuda")
    fn = lambda: [dec.decode(chunk) for _ in range(10)]
else:
    raise SystemExit("unknown mode " + mode)
with torch.no_grad():
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print("%s: %.3f ms per pass (%d passes after 2 warm-up passes)" % (mode, (time.time() - t) / reps * 1e3, reps))
