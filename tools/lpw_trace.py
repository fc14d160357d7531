"""Per-phase time of the launch-persistent forward kernel (csrc/stack_kernels.hip, stack_fwd_lpw_kernel) at the
E6D2 bench geometry: the first workgroup of every layer accumulates the 100 MHz ticks its lane 0 spent in
[wait for peers | h loads + MFMA | hand-off + cell | publish + drain | arrive | trailing stores].
usage: python tools/lpw_trace.py [STEPS] [CHUNK]      (run on the GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from edgedict_amd import _lib, encoder_stack  # noqa: E402
from edgedict_amd.models import Encoder  # noqa: E402

steps = sys.argv[1] if len(sys.argv) > 1 else "16"
encoder_stack.CHUNK = int(sys.argv[2]) if len(sys.argv) > 2 else 16
os.environ["EDGEDICT_STACK_LPW"] = "1"
os.environ["EDGEDICT_LPW_STEPS"] = steps
torch.manual_seed(0)
enc = Encoder(240, 1024, 6, 0.0, 640).cuda()
enc.compute_dtype = torch.bfloat16
xs = torch.randn(int(os.environ.get("EDGEDICT_TRACE_B", "64")), 401, 240, device="cuda")      # EDGEDICT_TRACE_B: rows of the batch
with torch.no_grad():
    for _ in range(3):
        enc(xs)
    torch.cuda.synchronize()
    buf = torch.zeros(8192, dtype=torch.int64, device="cuda")
    _lib.load().edgedict_stack_wsr_set_trace(_lib.ptr(buf))
    import time
    t0 = time.perf_counter()
    enc(xs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.load().edgedict_stack_wsr_set_trace(None)
tr = buf.cpu().view(-1, 8)[:6].double()
names = ["wait", "loads+mfma", "handoff+cell", "publish(+drain)", "arrive", "trailing"]
print("steps per launch %s, chunk %d, sub %s, poll %s, forward %.3f ms" % (steps, encoder_stack.CHUNK, os.environ.get("EDGEDICT_LPW_SUB", "-"), os.environ.get("EDGEDICT_LPW_POLL", "-"), dt * 1e3))
for l in range(6):
    n, launches = tr[l, 6].item(), tr[l, 7].item()
    if n == 0:
        continue
    per = [tr[l, i].item() / n * 0.01 for i in range(6)]
    print("layer %d: %4d steps in %3d launches; us per step: %s | sum %.2f" %
          (l, n, launches, "  ".join("%s %.2f" % (a, b) for a, b in zip(names, per)), sum(per)))
