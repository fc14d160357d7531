"""Per-step GPU time of the first 24 training steps (events at step boundaries): shows how long the
engine takes to reach steady state (allocator, pinned staging ring, vendor-library heuristics)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from edgedict_amd.flags import make_flags
from edgedict_amd.trainer import TrainEngine

flags = make_flags("E6D2", gradclip=None, dither=1e-5)
flags.sub_batch_size = 64
torch.manual_seed(0)
dev = torch.device("cuda", 0)
eng = TrainEngine(flags, device=dev, compute_dtype="bf16")
batch = bench.synth_batch(flags, 64, 15.0, 64, 1000, dev)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(25)]
host = []
torch.cuda.synchronize()
evs[0].record()
for i in range(24):
    t = time.perf_counter()
    eng.train_step(*batch)
    host.append(1e3 * (time.perf_counter() - t))
    evs[i + 1].record()
torch.cuda.synchronize()
print("gpu ms :", " ".join("%.1f" % evs[i].elapsed_time(evs[i + 1]) for i in range(24)))
print("host ms:", " ".join("%.1f" % h for h in host))
print("reserved GB %.1f, allocated GB %.1f, alloc retries %d, segments %d" % (
    torch.cuda.memory_reserved() / 1e9, torch.cuda.memory_allocated() / 1e9,
    torch.cuda.memory_stats()["num_alloc_retries"], torch.cuda.memory_stats()["segment.all.current"]))
