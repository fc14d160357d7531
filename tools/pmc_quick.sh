#!/bin/bash
# On the GPU box: per-kernel counter averages of a short command.
#   bash tools/pmc_quick.sh "<counters of one pass>" "<kernel name substring>" <command ...>
# (rocprofv3 --pmc with --kernel-trace only; one pass per call)
C=$1; K=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcq
timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmcq -o p -- "$@" > /tmp/pmcq.log 2>&1
DB=$(find /tmp/pmcq -name "*results.db" | head -1)
python - "$DB" "$K" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='view' or type='table'")]
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
kn = "kernel_name" if "kernel_name" in cols else "name"
cn = "counter_name" if "counter_name" in cols else "pmc_name"
vn = "value" if "value" in cols else "counter_value"
did = "dispatch_id" if "dispatch_id" in cols else "id"
rows = c.execute("select %s, %s, %s, sum(%s) from counters_collection group by %s, %s, %s" % (kn, did, cn, vn, kn, did, cn)).fetchall()
agg = {}
for name, d, ctr, v in rows:
    if sys.argv[2] not in name:
        continue
    a = agg.setdefault((name.split("(")[0][-60:], ctr), [0, 0.0])
    a[0] += 1; a[1] += v
for (name, ctr), (n, s) in sorted(agg.items()):
    print("%-60s %-28s launches %4d  avg %.4g" % (name, ctr, n, s / n))
PY
