"""Host position vs GPU position at marked points of a training step (steady state).
For every mark: host time since the step's first mark, and the time at which the GPU's CURRENT
stream reached the same point.  host << gpu: the host runs ahead (good); host ~ gpu: the GPU waits
for the host there."""
import os
import sys

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
eng = TrainEngine(flags, device=dev, compute_dtype="bf16")
batch = bench.synth_batch(flags, 64, 15.0, 64, 1000, dev)
for _ in range(3):
    eng.train_step(*batch)
ops.MARKS = []
for _ in range(3):
    eng.train_step(*batch)
torch.cuda.synchronize()
marks = ops.MARKS
ops.MARKS = None
starts = [i for i, m in enumerate(marks) if m[0] == "step:enter"]
i0 = starts[1]                     # second marked step: steady state
i1 = starts[2]
t0, e0 = marks[i0][1], marks[i0][2]
print("%-18s %10s %10s" % ("mark", "host ms", "gpu ms"))
for tag, t, ev in marks[i0:i1 + 1]:
    print("%-18s %10.2f %10.2f" % (tag, 1e3 * (t - t0), e0.elapsed_time(ev)))

# ---- the same step enqueued against an IDLE GPU (synchronised first): the host columns are then
# the un-throttled enqueue cost, the gpu column shows where the GPU has to wait for the host
torch.cuda.synchronize()
ops.MARKS = []
eng.train_step(*batch)
torch.cuda.synchronize()
marks = ops.MARKS
ops.MARKS = None
t0, e0 = marks[0][1], marks[0][2]
print("\nafter a device synchronise (GPU idle at step:enter)")
print("%-18s %10s %10s" % ("mark", "host ms", "gpu ms"))
for tag, t, ev in marks:
    print("%-18s %10.2f %10.2f" % (tag, 1e3 * (t - t0), e0.elapsed_time(ev)))
