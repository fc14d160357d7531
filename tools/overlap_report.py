"""Un-perturbed overlap evidence for one training step (E6D2 bench geometry), from in-kernel stamps and host/GPU
marks - no profiler attached (rocprofv3 --kernel-trace stretches the step by ~5 ms and re-orders the background
products, profiles/README.md):

  * step marks (tools/host_vs_gpu.py): where the GPU is when the host reaches each phase;
  * the recurrence launch sequences of the encoder stack (edgedict_stack_launch_stamps): span, sum of kernel
    durations, sum of gaps, mean duration by number of layers in the launch - forward (launch-persistent) and BPTT;
  * the same with each background product family moved out from under the BPTT:
      default                      joint dW2 + the stack's dW_ih/dW_hh/db on the auxiliary stream under the BPTT
      stack weight grads at end    EDGEDICT_STACK_FLAGS=2 (DW_AT_END): after the last BPTT launch
      nothing deferred             config.DEFER_WEIGHT_GRADS off: joint dW2 inline before the BPTT, stack grads returned
usage: python tools/overlap_report.py > profiles/r3_overlap.txt      (run on the GPU box)"""
import ctypes
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(variant):
    import numpy as np
    import torch
    import bench
    from edgedict_amd import _lib, config, ops
    from edgedict_amd.flags import make_flags
    from edgedict_amd.trainer import TrainEngine
    if variant == "nodefer":
        config.DEFER_WEIGHT_GRADS = False
    lib = _lib.load()
    flags = make_flags("E6D2", gradclip=None, dither=1e-5)
    flags.sub_batch_size = 64
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    eng = TrainEngine(flags, device=dev, compute_dtype="bf16")
    batch = bench.synth_batch(flags, 64, 15.0, 64, 1000, dev)
    nxt = None if variant == "serial front-end" else (batch[0], batch[1])      # bench.py prefetches the next front-end
    for _ in range(4):
        eng.train_step(*batch, next_batch=nxt)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(10):
        eng.train_step(*batch, next_batch=nxt)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100
    lib.edgedict_stack_time_launches(1)
    ops.MARKS = []
    for _ in range(2):
        eng.train_step(*batch, next_batch=nxt)
    torch.cuda.synchronize()
    marks, ops.MARKS = ops.MARKS, None
    print("== %s: %.2f ms per step (10 un-stamped steps)" % (variant, ms))
    starts = [i for i, m in enumerate(marks) if m[0] == "step:enter"]
    i0 = starts[-1]
    e0 = marks[i0][2]
    print("   marks (gpu ms since step:enter): " + "  ".join("%s %.2f" % (tag, e0.elapsed_time(ev)) for tag, t, ev in marks[i0:]))
    for name, bw in (("forward", 0), ("BPTT", 1)):
        buf = (ctypes.c_ulonglong * 8192)()
        n = ctypes.c_int(0)
        if lib.edgedict_stack_launch_stamps(bw, buf, 4096, ctypes.byref(n)) != 0:
            continue
        raw = np.array(buf[:2 * n.value], dtype=np.uint64).reshape(-1, 2)
        ok = (raw[:, 1] > raw[:, 0]) & (raw[:, 0] != np.uint64(0xffffffffffffffff))   # slots of launches that stamped
        st = raw[ok].astype(np.float64) * 0.01
        if len(st) < 2:
            continue
        d = st[:, 1] - st[:, 0]
        g = st[1:, 0] - st[:-1, 1]
        print("   %-8s %3d stamped launches: span %.2f ms, kernels %.2f ms, gaps %.2f ms; mean kernel %.1f us, mean gap %.1f us, "
              "period %.1f us" % (name, len(st), (st[-1, 1] - st[0, 0]) * 1e-3, d.sum() * 1e-3, g.sum() * 1e-3, d.mean(), g.mean(),
                                  (st[-1, 1] - st[0, 0]) / len(st)))
    lib.edgedict_stack_time_launches(0)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        print(__doc__.split("usage:")[0])
        for variant, env in (("default", {}), ("serial front-end", {}), ("stack weight grads at end", {"EDGEDICT_STACK_FLAGS": "2"}),
                             ("nodefer", {}), ("forward hand-off through arrival counters (EDGEDICT_LPW_POLL=0)", {"EDGEDICT_LPW_POLL": "0"}),
                             ("forward one launch per step (EDGEDICT_STACK_LPW=0)", {"EDGEDICT_STACK_LPW": "0"})):
            e = dict(os.environ)
            e.update(env)
            arg = "nodefer" if variant == "nodefer" else variant
            r = subprocess.run([sys.executable, os.path.abspath(__file__), arg], env=e, capture_output=True, text=True)
            print(r.stdout.strip() or r.stderr[-800:])
