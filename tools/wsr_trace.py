#!/usr/bin/env python
"""From a rocprofv3 kernel-trace DB: the weights-stationary forward of the LAST training step -
each wsr_fwd_kernel launch (start, duration, gap to the previous one) with the side-stream kernels
(chunk norm, input products) that ran between it and the next launch."""
import sqlite3
import sys


def main(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name,start,end,grid_x,queue_id from kernels order by start").fetchall()
    fb = [r[1] for r in rows if "fbank" in r[0]]
    t0 = fb[-1]
    step = [r for r in rows if r[1] >= t0]
    wsr = [r for r in step if "wsr_fwd" in r[0]]
    if not wsr:
        print("no wsr_fwd_kernel in the last step")
        return
    print("last step: %d wsr_fwd launches, first at %.3f ms, last ends %.3f ms after the front-end"
          % (len(wsr), (wsr[0][1] - t0) / 1e6, (wsr[-1][2] - t0) / 1e6))
    prev_end = None
    for i, r in enumerate(wsr):
        nxt = wsr[i + 1][1] if i + 1 < len(wsr) else r[2] + 400000
        side = [x for x in step if r[1] <= x[1] < nxt and "wsr_fwd" not in x[0]]
        fam = {}
        for x in side:
            n = x[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:24]
            a = fam.setdefault(n, [0, 0.0, 0])
            a[0] += 1
            a[1] += (x[2] - x[1]) / 1e3
            a[2] = max(a[2], (x[2] - r[1]) / 1e3)
        print("  launch %2d: start %8.1f us  dur %7.1f us  gap %6.1f us | %s"
              % (i, (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, 0.0 if prev_end is None else (r[1] - prev_end) / 1e3,
                 "  ".join("%s x%d %.0fus (ends +%.0f)" % (k, v[0], v[1], v[2]) for k, v in fam.items())))
        prev_end = r[2]


if __name__ == "__main__":
    main(sys.argv[1])
