"""Kernel timeline around the encoder stack's backward pass in a rocprofv3 results DB (profiles/collect.sh):
what runs before the first and after the last BPTT launch of the last training step.
  python tools/step_tail.py gpurun_out/<tag>_results.db"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end,queue_id,stream_id,grid_x from kernels order by start"))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:70]


idx = [i for i, r in enumerate(rows) if "stack_bwd_sk" in r[0]]
groups = []
for i in idx:                                    # a step's launches are < 2 ms apart
    if groups and rows[i][1] - rows[groups[-1][-1]][2] < 2e6:
        groups[-1].append(i)
    else:
        groups.append([i])
g = groups[int(sys.argv[2]) if len(sys.argv) > 2 else -1]
t0, t1 = rows[g[0]][1], rows[g[-1]][2]
print("BPTT: %d launches, span %.2f ms" % (len(g), (t1 - t0) / 1e6))
print("--- the 40 kernels before the first BPTT launch (us relative to its start, duration, queue, name, grid)")
for r in rows[max(0, g[0] - 40):g[0]]:
    print("%9.1f %8.1f q%s %s grid %d" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], short(r[0]), r[5]))
print("--- kernels after the last BPTT launch (us relative to its end)")
for r in rows[g[-1] + 1:g[-1] + 70]:
    print("%9.1f %8.1f q%s %s grid %d" % ((r[1] - t1) / 1e3, (r[2] - r[1]) / 1e3, r[3], short(r[0]), r[5]))
    if "dither" in r[0] or "fbank" in r[0]:
        break
