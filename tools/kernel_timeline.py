#!/usr/bin/env python
"""Every kernel of the LAST encoder-stack forward pass in a rocprofv3 kernel-trace DB (from the last
stack_input_norm_kernel to the next one / end): start (us), duration (us), queue, grid, name.
usage: python tools/kernel_timeline.py results.db [t_from_us t_to_us]"""
import sqlite3
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:40]


def main(db, lo=None, hi=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name,start,end,grid_x,queue_id from kernels order by start").fetchall()
    starts = [r[1] for r in rows if "stack_input_norm_kernel" in r[0]]
    t0 = starts[-1]
    sel = [r for r in rows if r[1] >= t0]
    qs = sorted({r[4] for r in sel})
    print("last forward: %d kernels, span %.1f us, queues %s" % (len(sel), (max(r[2] for r in sel) - t0) / 1e3, qs))
    for r in sel:
        s, d = (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3
        if lo is not None and (s + d < lo or s > hi):
            continue
        print("%9.1f %8.1f  q%-2d %7d  %s" % (s, d, qs.index(r[4]), r[3], short(r[0])))


if __name__ == "__main__":
    main(sys.argv[1], *(float(x) for x in sys.argv[2:4]))
