"""Does creating other HIP streams BEFORE the engine's internal ones (as RCCL / torch.distributed do
when the first collective runs before the first encoder-stack call) change the step time?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from edgedict_amd.flags import make_flags
from edgedict_amd.trainer import TrainEngine

n_pre = int(sys.argv[1])
use = len(sys.argv) > 2
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
if os.environ.get("ENGINE_STREAMS_FIRST") == "1":
    from edgedict_amd import side
    side.stream(dev)          # creates the library's three internal streams now
nccl = os.environ.get("WITH_NCCL") == "1"
if nccl:      # RCCL communicator + its streams, created by the first collective
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    dist.all_reduce(torch.zeros(1024, device=dev))
pre = [torch.cuda.Stream(device=dev) for _ in range(n_pre)]
for s in pre:
    with torch.cuda.stream(s):
        torch.zeros(16, device=dev)
flags = make_flags("E6D2", gradclip=None, dither=1e-5)
flags.sub_batch_size = 64
torch.manual_seed(0)
eng = TrainEngine(flags, device=dev, compute_dtype="bf16")
batch = bench.synth_batch(flags, 64, 15.0, 64, 1000, dev)
buf = torch.zeros(50 << 20, device=dev)
def step():
    eng.train_step(*batch)
    if nccl:
        dist.all_reduce(eng.flat.grad)
    if use and pre:      # something like a gradient exchange on the foreign stream after the step
        pre[0].wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(pre[0]):
            buf.add_(1.0)
        torch.cuda.current_stream().wait_stream(pre[0])
for _ in range(4):
    step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(12):
    step()
torch.cuda.synchronize()
print("pre-created streams %d%s: %.2f ms/step" % (n_pre, " (used)" if use else "", (time.perf_counter() - t) / 12 * 1e3))
