#!/bin/bash
# gpurun: kernel stats of the four decode workloads -> gpurun_out/<tag>_decode_<mode>.md
TAG=${1:-r4c}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for MODE in greedy_fp32 greedy_bf16 stream256 beam10; do
  rm -rf /tmp/dprof_$MODE
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/dprof_$MODE -o t -- python $ROOT/tools/decode_profile.py $MODE > $OUT/${TAG}_decode_$MODE.log 2>&1
  DB=$(find /tmp/dprof_$MODE -name "*results.db" | head -1)
  (cd $ROOT && python profiles/summarize.py $DB $OUT/${TAG}_decode_$MODE.md "$TAG decode $MODE: 5 passes (2 warm-up + 3)")
done
