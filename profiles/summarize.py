#!/usr/bin/env python
"""Turn a rocprofv3 ``*_results.db`` (kernel trace) into a per-kernel stats table (markdown).

    python profiles/summarize.py gpurun_out/prof_x/x_results.db profiles/r1_x_kernel_stats.md "title"
"""
import sqlite3
import sys


def main(db, out, title):
    c = sqlite3.connect(db)
    # GEMM kernels serve many shapes under one name: split them by grid size (= 256 x tiles)
    rows = c.execute(
        "select name || case when name like '%gemm%' then ' [tiles=' || (grid_x / 256) || "
        "(case when grid_z > 1 then 'x' || grid_z else '' end) || ']' else '' end as nm, "
        "count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, "
        "max(end-start)/1e3, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size) "
        "from kernels group by nm order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(out, "w") as f:
        f.write("# %s\n\nSource: `rocprofv3 --kernel-trace --stats` (sqlite output), summed over the "
                "whole process.\nTotal kernel time: %.2f ms\n\n" % (title, tot))
        f.write("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | sgpr | lds |\n")
        f.write("|---|---|---|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            if r[2] / tot < 0.0005:
                continue
            name = r[0].replace("(anonymous namespace)::", "").replace("void ", "")
            tag = name[name.index(" [tiles="):] if " [tiles=" in name else ""
            name = name.split("(")[0][:80] + tag
            f.write("| `%s` | %d | %.2f | %.2f | %.2f | %.2f | %.1f | %s | %s | %s | %s |\n"
                    % (name, r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot, r[6], r[7], r[8], r[9]))
    print("wrote", out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "kernel stats")
