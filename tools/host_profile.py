"""Where does the HOST time of a training step go?  cProfile over a few steps (GPU box)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
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
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    eng.train_step(*batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
