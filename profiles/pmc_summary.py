#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

    python profiles/pmc_summary.py out.json fetch_results.db write_results.db [mfma_results.db]

Units and gfx950 corrections follow MI355X_MICROARCH.md (HBM section): both counters are in KiB
(bytes = value * 1024); on gfx950 FETCH_SIZE reports exactly HALF of the bytes of a wide coalesced
streaming read (128-byte requests tallied at 64 B), so `fetch_bytes_corrected` = 2 x raw is what
applies to 16-byte-per-lane streaming kernels (the GEMM operand loads, the loss kernels); WRITE_SIZE
is uncalibrated and reported raw.  Values are averages PER LAUNCH of each kernel name.
"""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='view' or type='table'")]
    view = "counters_collection" if "counters_collection" in tabs else None
    if view is None:
        raise SystemExit("no counters_collection view in %s" % db)
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % view)]
    kn = "kernel_name" if "kernel_name" in cols else "name"
    cn = "counter_name" if "counter_name" in cols else "pmc_name"
    vn = "value" if "value" in cols else "counter_value"
    did = "dispatch_id" if "dispatch_id" in cols else "id"
    gcol = next((x for x in ("grid_size_x", "grid_x", "grid_size") if x in cols), None)
    rows = c.execute("select %s, %s, sum(%s), %s from %s where %s = ? group by %s, %s"
                     % (kn, did, vn, ("max(%s)" % gcol) if gcol else "0", view, cn, kn, did),
                     (counter,)).fetchall()
    out = {}
    for name, _, v, grid in rows:
        name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "gemm" in name and grid:      # one kernel name, many shapes: split by tile count
            name += " [tiles=%d]" % (int(grid) // 256)
        a = out.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += float(v)
    return {k: (n, s / n) for k, (n, s) in out.items()}


def main(out, fdb, wdb, mdb=None):
    f = per_kernel(fdb, "FETCH_SIZE")
    w = per_kernel(wdb, "WRITE_SIZE")
    mfma = gui = {}
    if mdb:   # third pass: matrix-pipe busy cycles of all 1024 SIMDs over chip-active cycles.
              # rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCCs (a 1.34 ms kernel reports 22.6 M =
              # 8 x 2.1 GHz x 1.34 ms), so one XCC's active cycles are GUI_ACTIVE / 8:
              # util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)
        mfma = per_kernel(mdb, "SQ_VALU_MFMA_BUSY_CYCLES")
        gui = per_kernel(mdb, "GRBM_GUI_ACTIVE")
    res = {}
    for k in sorted(set(f) | set(w)):
        fn, fv = f.get(k, (0, 0.0))
        wn, wv = w.get(k, (0, 0.0))
        res[k] = {"launches": max(fn, wn),
                  "fetch_bytes_raw": fv * 1024.0, "fetch_bytes_corrected": 2.0 * fv * 1024.0,
                  "write_bytes_raw": wv * 1024.0,
                  "hbm_bytes": 2.0 * fv * 1024.0 + wv * 1024.0}
        if k in mfma and k in gui and gui[k][1] > 0:
            res[k]["mfma_busy_cycles"] = mfma[k][1]
            res[k]["gui_active_cycles"] = gui[k][1]
            res[k]["mfma_util"] = mfma[k][1] / (gui[k][1] / 8.0 * 256 * 4)
    with open(out, "w") as fh:
        json.dump({"note": "per-launch averages; FETCH_SIZE x2 gfx950 correction applied in "
                           "fetch_bytes_corrected / hbm_bytes (MI355X_MICROARCH.md, HBM section)",
                   "kernels": res}, fh, indent=1, sort_keys=True)
    top = sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes"] * kv[1]["launches"])[:12]
    for k, v in top:
        print("%-60s n=%5d  fetch(corr) %8.1f MB  write %8.1f MB  mfma_util %s" %
              (k[:60], v["launches"], v["fetch_bytes_corrected"] / 1e6, v["write_bytes_raw"] / 1e6,
               ("%.1f%%" % (100 * v["mfma_util"])) if "mfma_util" in v else "-"))


if __name__ == "__main__":
    main(*sys.argv[1:5])
