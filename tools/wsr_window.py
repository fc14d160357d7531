#!/usr/bin/env python
"""Raw kernel rows (queue, start, duration) of a window of the last step's weights-stationary forward:
python tools/wsr_window.py results.db [first_launch=8] [n_launches=2]"""
import sqlite3
import sys


def main(db, first=8, n=2):
    c = sqlite3.connect(db)
    rows = c.execute("select name,start,end,grid_x,queue_id from kernels order by start").fetchall()
    t0 = [r[1] for r in rows if "fbank" in r[0]][-1]
    step = [r for r in rows if r[1] >= t0]
    wsr = [r for r in step if "wsr_fwd" in r[0]]
    lo, hi = wsr[first][1], wsr[min(len(wsr) - 1, first + n)][1]
    for r in step:
        if lo <= r[1] < hi or (r[1] < lo < r[2]):
            name = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
            print("q%d  +%8.1f us  dur %7.1f us  grid %6d  %s" % (r[4], (r[1] - lo) / 1e3, (r[2] - r[1]) / 1e3, r[3], name))


if __name__ == "__main__":
    a = sys.argv
    main(a[1], int(a[2]) if len(a) > 2 else 8, int(a[3]) if len(a) > 3 else 2)
