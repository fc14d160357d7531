#!/bin/bash
# Run ON THE GPU BOX (gpurun) from the repo root:  bash profiles/collect.sh <tag>
# 1. rocprofv3 --kernel-trace --stats of `bench.py --steps 3 --warmup 1`  -> gpurun_out/<tag>_kernel_stats.md
# 2. two separate PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass: TCC has 4 slots, they cost
#    3 + 2) with --kernel-trace only, and a third pass for SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE
#                                                                            -> gpurun_out/<tag>_pmc.json
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-loss-delta --no-own-kernels-run --no-fp32-run --no-secondary"
rm -rf /tmp/prof_$TAG
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/trace -o t -- $CMD > $OUT/${TAG}_trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_$TAG/$C -o p -- $CMD > $OUT/${TAG}_pmc_$C.log 2>&1
done
# 3. matrix-pipe utilisation: SQ + GRBM counters share a pass
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/prof_$TAG/MFMA -o p -- $CMD > $OUT/${TAG}_pmc_MFMA.log 2>&1
cd $ROOT
DB=$(find /tmp/prof_$TAG/trace -name "*results.db" | head -1)
python profiles/summarize.py $DB $OUT/${TAG}_kernel_stats.md "$TAG: bench.py --steps 3 --warmup 1 (E6D2, bf16, B=64, 15 s): every training step of the run: warm-up, timed, the 2 stamped steps of the roofline measurement, the 3 steps of host_unthrottled_ms"
python profiles/pmc_summary.py $OUT/${TAG}_pmc.json $(find /tmp/prof_$TAG/FETCH_SIZE -name "*results.db" | head -1) $(find /tmp/prof_$TAG/WRITE_SIZE -name "*results.db" | head -1) $(find /tmp/prof_$TAG/MFMA -name "*results.db" | head -1)
cp $DB $OUT/${TAG}_results.db
