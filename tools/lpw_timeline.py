"""Launch timeline of the encoder stack's forward pass at the E6D2 bench geometry: per wavefront launch its
start / duration / gap to the previous launch (in-kernel stamps) and the layers it carried (dry-run schedule).
usage: python tools/lpw_timeline.py LPW STEPS CHUNK [MARGIN]     (run on the GPU box)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from edgedict_amd import _lib, encoder_stack  # noqa: E402
from edgedict_amd.models import Encoder  # noqa: E402

os.environ["EDGEDICT_STACK_LPW"] = sys.argv[1] if len(sys.argv) > 1 else "1"
os.environ["EDGEDICT_LPW_STEPS"] = sys.argv[2] if len(sys.argv) > 2 else "6"
encoder_stack.CHUNK = int(sys.argv[3]) if len(sys.argv) > 3 else 12
if len(sys.argv) > 4:
    os.environ["EDGEDICT_LPW_MARGIN"] = sys.argv[4]
lib = _lib.load()
torch.manual_seed(0)
enc = Encoder(240, 1024, 6, 0.0, 640).cuda()
enc.compute_dtype = torch.bfloat16
xs = torch.randn(64, 401, 240, device="cuda")
with torch.no_grad():
    for _ in range(3):
        enc(xs)
    torch.cuda.synchronize()
    lib.edgedict_stack_time_launches(1)
    enc(xs)
    torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8192)()
n = ctypes.c_int(0)
assert lib.edgedict_stack_launch_stamps(0, buf, 4096, ctypes.byref(n)) == 0
lib.edgedict_stack_time_launches(0)
st = np.array(buf[:2 * n.value], dtype=np.float64).reshape(-1, 2) * 0.01      # us
steps, enq, nl, ms = encoder_stack.schedule(401, 240, 1024, [1, 2, 1, 1, 1, 1], B=64, chunk=encoder_stack.CHUNK)
per_launch = [[0] * 6 for _ in range(nl)]
for l in range(6):
    for t, w in enumerate(steps[l]):
        per_launch[w][l] += 1
t0 = st[0, 0]
print("launches stamped %d, scheduled %d; span %.1f us; sum of kernel durations %.1f us" %
      (n.value, nl, st[-1, 1] - t0, (st[:, 1] - st[:, 0]).sum()))
print("  k   start    dur    gap   steps per layer")
for k in range(n.value):
    gap = st[k, 0] - st[k - 1, 1] if k else 0.0
    if k < 40 or k % 10 == 0 or k > n.value - 15:
        print("%3d %7.1f %6.1f %6.1f   %s" % (k, st[k, 0] - t0, st[k, 1] - st[k, 0], gap,
                                              per_launch[k] if k < nl else "?"))
d = st[:, 1] - st[:, 0]
g = st[1:, 0] - st[:-1, 1]
print("mean duration %.1f us, mean gap %.1f us, max gap %.1f us" % (d.mean(), g.mean(), g.max()))
