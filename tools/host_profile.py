"""Where does the HOST time of a training step go?  cProfile (GPU box) over (a) 5 steps that start on an idle device:
the un-throttled enqueue cost by function; (b) 30 steps in steady state, where the host runs a queue ahead of the GPU
and BLOCKS wherever a queue is full: the functions with the largest own time are the throttle points.
(cProfile is imported after torch: imported before, it made every step slower - tools/README.md.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from edgedict_amd.flags import make_flags  # noqa: E402
from edgedict_amd.trainer import TrainEngine  # noqa: E402
import cProfile  # noqa: E402
import pstats  # noqa: E402
import time  # noqa: E402

flags = make_flags("E6D2", gradclip=None, dither=1e-5)
flags.sub_batch_size = 64
torch.manual_seed(0)
dev = torch.device("cuda", 0)
eng = TrainEngine(flags, device=dev, compute_dtype="bf16")
batch = bench.synth_batch(flags, 64, 15.0, 64, 1000, dev)
for _ in range(3):
    eng.train_step(*batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    eng.train_step(*batch)
pr.disable()
torch.cuda.synchronize()
print("== (a) 5 steps from an idle device: cumulative")
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
for _ in range(10):
    eng.train_step(*batch)
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
for _ in range(30):
    eng.train_step(*batch)
pr.disable()
th = time.time() - t0
torch.cuda.synchronize()
print("== (b) 30 steps in steady state: host %.2f ms per step, device %.2f ms per step; by OWN time" % (th / 30 * 1e3, (time.time() - t0) / 30 * 1e3))
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
