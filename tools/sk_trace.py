"""Per-phase time of the split-K BPTT kernel (csrc/stack_kernels.hip, stack_bwd_sk_kernel) at the E6D2 bench
geometry: workgroup 0 of every layer accumulates the 100 MHz ticks its lane 0 spent in
[wait for the layer | dG ring + MFMA | partial stores + unit-block wait | partial reads + cell | publish + drain |
trailing stores].
usage: python tools/sk_trace.py [STEPS]      (run on the GPU box)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["EDGEDICT_STACK_BWD_SK"] = "1"
os.environ["EDGEDICT_SK_STEPS"] = sys.argv[1] if len(sys.argv) > 1 else "16"
import torch  # noqa: E402

from edgedict_amd import _lib  # noqa: E402
from edgedict_amd.models import Encoder  # noqa: E402

lib = _lib.load()
torch.manual_seed(0)
enc = Encoder(240, 1024, 6, 0.0, 640).cuda()
enc.compute_dtype = torch.bfloat16
xs = torch.randn(int(os.environ.get("EDGEDICT_TRACE_B", "64")), 401, 240, device="cuda")      # EDGEDICT_TRACE_B: rows of the batch
for p in enc.parameters():
    p.grad = torch.zeros_like(p)
buf = torch.zeros(8192, dtype=torch.int64, device="cuda")
for it in range(4):
    out, _ = enc(xs)
    loss = out.float().sum()
    torch.cuda.synchronize()
    if it == 3:
        lib.edgedict_stack_wsr_set_trace(_lib.ptr(buf))
    t0 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
lib.edgedict_stack_wsr_set_trace(None)
tr = buf.cpu().view(-1, 8)[:6].double()
names = ["wait", "ring+mfma", "partial+blockwait", "reads+cell", "publish+drain", "trailing"]
print("steps per launch %s, backward %.3f ms (traced)" % (os.environ["EDGEDICT_SK_STEPS"], dt * 1e3))
for l in range(6):
    n, launches = tr[l, 6].item(), tr[l, 7].item()
    if n == 0:
        continue
    per = [tr[l, i].item() / n * 0.01 for i in range(6)]
    print("layer %d: %4d steps in %3d launches; us per step: %s | sum %.2f" %
          (l, n, launches, "  ".join("%s %.2f" % (a, b) for a, b in zip(names, per)), sum(per)))
