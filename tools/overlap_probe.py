#!/usr/bin/env python
"""What does work on a FIFTH stream cost while the backward pass runs?  (the overlapped gradient
exchange puts RCCL's kernels on a stream of their own beside the BPTT, and HIP has four hardware queues.)
One GPU, no communicator: dp.READY_HOOK is pointed at a stand-in that, like BucketedAllReduce.ready(),
orders a foreign stream behind the auxiliary stream's position and runs `passes` read-modify-write passes
over the layer's gradient slice there (34 MB per encoder layer; a ring all-reduce step moves comparable
bytes per bucket, slower).  Prints ms per training step without / with the stand-in."""
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (flag presets, synthetic batch)
from edgedict_amd import dp  # noqa: E402
from edgedict_amd.trainer import TrainEngine  # noqa: E402


def nccl_main():
    """One-rank RCCL communicator, EDGEDICT_DP_FORCE=1: the real BucketedAllReduce issue path (hooks, ready(),
    finish()) with the process group's stream handling, in the mode EDGEDICT_DP_EARLY / EDGEDICT_DP_OVERLAP
    select.  The collective itself is trivial on one rank; what is measured is what its STREAM costs."""
    import torch.distributed as dist
    from edgedict_amd import side
    from edgedict_amd.flags import make_flags
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29651")
    os.environ.setdefault("EDGEDICT_DP_FORCE", "1")
    torch.cuda.set_device(0)
    side.stream(torch.device("cuda", 0))
    dist.init_process_group("nccl", rank=0, world_size=1)
    flags = make_flags("E6D2", gradclip=None, dither=1e-5)
    flags.preset_name = "E6D2"
    flags.sub_batch_size = 64
    torch.manual_seed(0)
    eng = TrainEngine(flags, device=torch.device("cuda", 0), compute_dtype="bf16")
    batch = bench.synth_batch(flags, 64, 15.0, 64, 1000, torch.device("cuda", 0))
    for _ in range(4):
        eng.train_step(*batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 12
    for _ in range(K):
        eng.train_step(*batch)
    torch.cuda.synchronize()
    red = eng.reducer
    print("one-rank RCCL, overlap=%s early=%s: %.2f ms per step; buckets %d, left during backward %d (%s)"
          % (os.environ.get("EDGEDICT_DP_OVERLAP", "1"), red.early_mode, 1e3 * (time.perf_counter() - t0) / K,
             len(red.bounds), red.last_issued_early, ",".join(red.last_early_by)), flush=True)
    dist.destroy_process_group()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "nccl":
        return nccl_main()
    passes = [int(a) for a in sys.argv[1:]] or [0, 1, 8, 32]
    torch.cuda.set_device(0)
    from edgedict_amd import side
    from edgedict_amd.flags import make_flags
    side.stream(torch.device("cuda", 0))       # the engine's streams first
    flags = make_flags("E6D2", gradclip=None, dither=1e-5)
    flags.preset_name = "E6D2"
    flags.sub_batch_size = 64
    torch.manual_seed(0)
    eng = TrainEngine(flags, device=torch.device("cuda", 0), compute_dtype="bf16")
    batch = bench.synth_batch(flags, 64, 15.0, 64, 1000, torch.device("cuda", 0))
    foreign = torch.cuda.Stream()          # created AFTER the engine's streams, as RCCL's are
    red = eng.reducer
    for n in passes:
        def hook(params, stream=None, n=n):
            if n == 0:
                return
            ev = torch.cuda.Event()
            ev.record(stream if stream is not None else torch.cuda.current_stream())
            foreign.wait_event(ev)
            with torch.cuda.stream(foreign):
                for p in params:
                    b = red.param_bucket.get(id(p))
                    if b is None or getattr(hook, "seen", None) == (b, hook.step):
                        continue
                    hook.seen = (b, hook.step)
                    lo, hi = red.bounds[b]
                    g = eng.flat.grad[lo:hi]
                    for _ in range(n):
                        g.mul_(1.0)
        hook.step = 0
        dp.READY_HOOK = hook
        for _ in range(3):
            hook.step += 1
            eng.train_step(*batch)
            torch.cuda.current_stream().wait_stream(foreign)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 12
        for _ in range(K):
            hook.step += 1
            eng.train_step(*batch)
            torch.cuda.current_stream().wait_stream(foreign)
        torch.cuda.synchronize()
        print("passes %3d over each layer's gradient slice on a foreign stream: %.2f ms per step"
              % (n, 1e3 * (time.perf_counter() - t0) / K), flush=True)
    dp.READY_HOOK = None


if __name__ == "__main__":
    main()
