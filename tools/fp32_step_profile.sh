#!/bin/bash
# Run ON THE GPU BOX: kernel table of the token-exact fp32 training step (E6D2, B=64, 15 s).
#   bash tools/fp32_step_profile.sh <tag>   -> gpurun_out/<tag>_fp32_step_kernels.md
set -u
TAG=${1:-r6}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --dtype fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-loss-delta --no-own-kernels-run --no-fp32-run --no-secondary"
rm -rf /tmp/prof_fp32_$TAG
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_fp32_$TAG/trace -o t -- $CMD > $OUT/${TAG}_fp32_trace.log 2>&1
cd $ROOT
DB=$(find /tmp/prof_fp32_$TAG/trace -name "*results.db" | head -1)
python profiles/summarize.py $DB $OUT/${TAG}_fp32_step_kernels.md "$TAG: bench.py --dtype fp32 --steps 3 --warmup 1 (E6D2, token-exact fp32 mode, B=64, 15 s)"
tail -3 $OUT/${TAG}_fp32_trace.log
