"""Phase timeline of the token-exact fp32 training step (E6D2 bench geometry): GPU time at each step mark plus the
event-timed kernel families (ops.timed).   python tools/fp32_step_marks.py     (run on the GPU box)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from edgedict_amd import ops  # noqa: E402
from edgedict_amd.flags import make_flags  # noqa: E402
from edgedict_amd.trainer import TrainEngine  # noqa: E402

flags = make_flags("E6D2", gradclip=None, dither=1e-5)
flags.sub_batch_size = 64
torch.manual_seed(0)
dev = torch.device("cuda", 0)
eng = TrainEngine(flags, device=dev, compute_dtype="fp32")
batch = bench.synth_batch(flags, 64, 15.0, 64, 1000, dev)
nxt = (batch[0], batch[1])
for _ in range(2):
    eng.train_step(*batch, next_batch=nxt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4):
    eng.train_step(*batch, next_batch=nxt)
torch.cuda.synchronize()
print("fp32 step: %.2f ms (4 steps)" % ((time.perf_counter() - t0) * 250))
ops.MARKS = []
ops.TIMERS = {}
for _ in range(2):
    eng.train_step(*batch, next_batch=nxt)
torch.cuda.synchronize()
marks, ops.MARKS = ops.MARKS, None
starts = [i for i, m in enumerate(marks) if m[0] == "step:enter"]
i0 = starts[-1]
e0 = marks[i0][2]
h0 = marks[i0][1]
print("marks: tag  gpu_ms  host_ms (since step:enter)")
for tag, t, ev in marks[i0:]:
    print("   %-18s %8.2f %8.2f" % (tag, e0.elapsed_time(ev), 1e3 * (t - h0)))
for k, v in sorted(ops.timer_summary().items()):
    print("   timed %-24s %8.3f ms per step" % (k, v[1]))
