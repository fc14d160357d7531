#!/usr/bin/env python
"""What does a collective library's resident kernel cost the default recurrence kernels, and can it starve them?

No multi-GPU box is available to the builder, so RCCL's footprint is stood in for: at every point where
the gradient exchange would issue a bucket from inside the backward pass (dp.READY_HOOK: the joint's block and one call
per encoder layer, `edgedict_stack_desc_t.grads_final`), `edgedict_debug_footprint` runs on the AUXILIARY stream - where
the stream-mode exchange issues its collectives - with N workgroups x 512 threads x >= 128 registers per lane streaming
that layer's gradient slice (34 MB per encoder layer) `passes` times.  The launch-persistent forward (one workgroup per
CU, 361 registers) and the split-K BPTT (253 registers, 2 per CU) need all their workgroups co-resident; prints ms per
training step for each N and asserts that no bounded in-kernel wait gave up.

usage: python tools/rccl_footprint.py [N ...]          (default: 0 16 32 64; run on the GPU box)
       python tools/rccl_footprint.py nccl             (a ONE-rank RCCL communicator through the real issue path)"""
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (flag presets, synthetic batch)
from edgedict_amd import _lib, dp, encoder_stack, side  # noqa: E402
from edgedict_amd.flags import make_flags  # noqa: E402
from edgedict_amd.trainer import TrainEngine  # noqa: E402


def build(batch=64, seconds=15.0, labels=64):
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    side.stream(dev)                         # the engine's streams first
    flags = make_flags("E6D2", gradclip=None, dither=1e-5)
    flags.preset_name = "E6D2"
    flags.sub_batch_size = batch
    torch.manual_seed(0)
    eng = TrainEngine(flags, device=dev, compute_dtype="bf16")
    data = bench.synth_batch(flags, batch, seconds, labels, 1000, dev)
    return eng, data


def footprint_hook(eng, workgroups, passes):
    lib = _lib.load()
    red = eng.reducer
    state = {"step": 0, "seen": None, "calls": 0}

    def hook(params, stream=None):
        if workgroups == 0:
            return
        st = stream if stream is not None else torch.cuda.current_stream()
        for p in params:
            b = red.param_bucket.get(id(p))
            if b is None or state["seen"] == (b, state["step"]):
                continue
            state["seen"] = (b, state["step"])
            lo, hi = red.bounds[b]
            g = eng.flat.grad[lo:hi]
            state["calls"] += 1
            _lib.check(lib.edgedict_debug_footprint(ctypes.c_void_p(g.data_ptr()), ctypes.c_longlong(g.numel()),
                                                    workgroups, passes, ctypes.c_void_p(st.cuda_stream)),
                       "debug_footprint")
    return hook, state


def measure(eng, data, workgroups, passes=2, steps=10):
    hook, state = footprint_hook(eng, workgroups, passes)
    dp.READY_HOOK = hook if workgroups else None
    try:
        for _ in range(3):
            state["step"] += 1
            eng.train_step(*data)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            state["step"] += 1
            eng.train_step(*data)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
    finally:
        dp.READY_HOOK = None
    encoder_stack.check_wsr_error()          # raises if a bounded wait gave up
    return ms, state["calls"], encoder_stack.last_mode(False), encoder_stack.last_mode(True)


def nccl_main():
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29653")
    os.environ["EDGEDICT_DP_FORCE"] = "1"
    torch.cuda.set_device(0)
    side.stream(torch.device("cuda", 0))
    dist.init_process_group("nccl", rank=0, world_size=1)
    eng, data = build()
    for _ in range(3):
        eng.train_step(*data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        eng.train_step(*data)
    torch.cuda.synchronize()
    encoder_stack.check_wsr_error()
    red = eng.reducer
    print(json.dumps({"one_rank_rccl_ms_per_step": 1e3 * (time.perf_counter() - t0) / 10,
                      "left_during_backward": red.last_issued_early, "buckets": len(red.bounds),
                      "fwd_mode": encoder_stack.last_mode(False), "bwd_mode": encoder_stack.last_mode(True)}), flush=True)
    dist.destroy_process_group()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "nccl":
        return nccl_main()
    ns = [int(a) for a in sys.argv[1:]] or [0, 16, 32, 64]
    eng, data = build()
    rows = []
    for n in ns:
        ms, calls, fm, bm = measure(eng, data, n)
        rows.append({"workgroups": n, "ms_per_step": round(ms, 3), "footprint_launches_per_step": calls // 13 if n else 0,
                     "fwd_mode": fm, "bwd_mode": bm, "gave_up": False})
        print(json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
