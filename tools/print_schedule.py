"""Print the dry-run wavefront schedule of the E6D2 encoder stack (forward; `b` = backward): frames per layer and launch."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgedict_amd import encoder_stack as es
red=[1,2,1,1,1,1]
bwd = len(sys.argv) > 1 and sys.argv[1] == "b"
steps, enq, n, ms = es.schedule(401, 240, 1024, red, B=64, chunk=16, backward=bwd)
print("backward" if bwd else "forward", "launches", n, "max_slots", ms)
for w in range(n):
    row=[]
    for l in range(6):
        ts=np.nonzero(steps[l]==w)[0]
        row.append("%3d-%3d"%(ts[0],ts[-1]) if len(ts) else "   .   ")
    print("%2d: "%w+"  ".join(row))
