"""In-kernel timeline of the persistent weights-stationary forward (E6D2 bench batch): per layer and chunk
when its CU 0 started waiting for the input product, started computing, and finished; per task of worker 0
when it started waiting for the frames, finished waiting, started the product, finished."""
import os
import sys

os.environ["EDGEDICT_STACK_WSR"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from edgedict_amd import _lib  # noqa: E402
from edgedict_amd.flags import make_flags  # noqa: E402
from edgedict_amd.trainer import TrainEngine  # noqa: E402

flags = make_flags("E6D2", gradclip=None, dither=1e-5)
flags.sub_batch_size = 64
torch.manual_seed(0)
dev = torch.device("cuda", 0)
eng = TrainEngine(flags, device=dev, compute_dtype="bf16")
batch = bench.synth_batch(flags, 64, 15.0, 64, 1000, dev)
for _ in range(3):
    eng.train_step(*batch)
torch.cuda.synchronize()
buf = torch.zeros(8192, dtype=torch.int64, device=dev)
_lib.load().edgedict_stack_wsr_set_trace(_lib.ptr(buf))
eng.train_step(*batch)
torch.cuda.synchronize()
_lib.load().edgedict_stack_wsr_set_trace(None)
t = buf.cpu().numpy()
base = min(x for x in t[:2048 + 6 * 64 * 4] if x > 0)
us = lambda x: (x - base) / 100.0 if x > 0 else float("nan")
print("layer chunk   wait_start  compute_start  chunk_end   (us since the first stamp)")
for l in range(6):
    for k in range(0, 17):
        a = t[(l * 64 + k) * 4:(l * 64 + k) * 4 + 3]
        if a[1] > 0:
            print("%5d %5d %12.1f %13.1f %10.1f   wait %.1f compute %.1f" % (l, k, us(a[0]), us(a[1]), us(a[2]),
                                                                              us(a[1]) - us(a[0]), us(a[2]) - us(a[1])))
print("worker 0:  layer chunk  wait_start  frames_out  product_start  done")
for l in range(6):
    for k in range(0, 17):
        a = t[2048 + (l * 64 + k) * 4:2048 + (l * 64 + k) * 4 + 4]
        if a[0] > 0:
            print("           %5d %5d %10.1f %11.1f %14.1f %8.1f   ln %.1f gemm %.1f" % (
                l, k, us(a[0]), us(a[1]), us(a[2]), us(a[3]), us(a[2]) - us(a[1]), us(a[3]) - us(a[2])))
