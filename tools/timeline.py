#!/usr/bin/env python
"""Timeline digest of a rocprofv3 kernel-trace DB: for the LAST training step (from the last
fbank_kernel on), per-queue busy time, and a coarse Gantt (kernel families per 1 ms bucket)."""
import collections
import sqlite3
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0]
    if n.startswith("gemm_kernel"):
        n = "gemm" + ("_f32out" if "float" in n.split(",")[1] else "") + ("_" + "".join("T" if "true" in x else "N" for x in n.split(",")[2:4]))
    return n[:28]


def main(db, bucket_ms=1.0):
    c = sqlite3.connect(db)
    rows = c.execute("select name,start,end,grid_x,queue_id from kernels order by start").fetchall()
    fb = [r[1] for r in rows if "fbank" in r[0]]
    t0 = fb[-1]
    step = [r for r in rows if r[1] >= t0]
    t1 = max(r[2] for r in step)
    print("last step: %d kernels, span %.2f ms" % (len(step), (t1 - t0) / 1e6))
    byq = collections.defaultdict(float)
    for r in step:
        byq[r[4]] += (r[2] - r[1]) / 1e6
    print("busy ms per queue:", {q: round(v, 2) for q, v in sorted(byq.items())})
    nb = int((t1 - t0) / 1e6 / bucket_ms) + 1
    g = [collections.defaultdict(float) for _ in range(nb)]
    for r in step:
        s, e = (r[1] - t0) / 1e6, (r[2] - t0) / 1e6
        b = int(s / bucket_ms)
        while b < nb and b * bucket_ms < e:
            lo, hi = max(s, b * bucket_ms), min(e, (b + 1) * bucket_ms)
            if hi > lo:
                g[b]["q%d:%s" % (r[4], short(r[0]))] += hi - lo
            b += 1
    for b in range(nb):
        top = sorted(g[b].items(), key=lambda kv: -kv[1])[:4]
        print("%5.1f ms | " % (b * bucket_ms) + "  ".join("%s %.2f" % kv for kv in top))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
