"""Launch timeline of the encoder stack's BACKWARD pass at the E6D2 bench geometry (in-kernel stamps of every
BPTT launch + the dry-run schedule): duration / gap statistics by number of layer-steps in the launch.
usage: python tools/bwd_timeline.py      (run on the GPU box)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from edgedict_amd import _lib, encoder_stack  # noqa: E402
from edgedict_amd.models import Encoder  # noqa: E402

lib = _lib.load()
torch.manual_seed(0)
enc = Encoder(240, 1024, 6, 0.0, 640).cuda()
enc.compute_dtype = torch.bfloat16
xs = torch.randn(64, 401, 240, device="cuda")
for p in enc.parameters():
    p.grad = torch.zeros_like(p)
for it in range(4):
    if it == 3:
        lib.edgedict_stack_time_launches(1)
    out, _ = enc(xs)
    out.float().sum().backward()
    torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8192)()
n = ctypes.c_int(0)
assert lib.edgedict_stack_launch_stamps(1, buf, 4096, ctypes.byref(n)) == 0
lib.edgedict_stack_time_launches(0)
st = np.array(buf[:2 * n.value], dtype=np.float64).reshape(-1, 2) * 0.01
steps, enq, nl, ms = encoder_stack.schedule(401, 240, 1024, [1, 2, 1, 1, 1, 1], B=64, chunk=encoder_stack.CHUNK, backward=True)
cnt = np.zeros(nl, dtype=int)
for l in range(6):
    np.add.at(cnt, steps[l].astype(int), 1)
d = st[:, 1] - st[:, 0]
g = np.concatenate([[0.0], st[1:, 0] - st[:-1, 1]])
print("launches %d (scheduled %d); span %.1f us; kernel sum %.1f us; gap sum %.1f us" % (n.value, nl, st[-1, 1] - st[0, 0], d.sum(), g.sum()))
for k in sorted(set(cnt[:n.value])):
    m = cnt[:n.value] == k
    print("  %d layer-steps: %3d launches, mean duration %.1f us, mean gap before %.1f us" % (k, m.sum(), d[m].mean(), g[m].mean()))
big = np.argsort(-g)[:10]
print("largest gaps:", [(int(i), round(float(g[i]), 1)) for i in sorted(big)])
if os.environ.get("EDGEDICT_STACK_BWD_SK", "1") == "1":
    per = [[0] * 6 for _ in range(nl)]
    for l in range(6):
        for t, w in enumerate(steps[l]):
            per[w][l] += 1
    t0 = st[0, 0]
    for k in range(n.value):
        if k < 45 or k % 10 == 0:
            print("%3d %8.1f %6.1f %6.1f  %s" % (k, st[k, 0] - t0, d[k], g[k], per[k]))
