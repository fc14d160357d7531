#!/bin/bash
# round 4, first GPU pass: the sub-batched recurrence kernels - parity, then timings
mkdir -p gpurun_out/r4a
cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_lpw_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/r4a/test_lpw.txt
echo "rc=$?" >> gpurun_out/r4a/test_lpw.txt
timeout 300 python tools/fwd_time.py "BWD=1,SUB=1,SKSUB=1" "BWD=1,SUB=2,SKSUB=1" "BWD=1,SUB=4,SKSUB=1" "BWD=1,SUB=2,SKSUB=2" "BWD=1,SUB=2,SKSUB=4" "BWD=1,SUB=4,SKSUB=4" > gpurun_out/r4a/fwd_time.txt 2>&1
for cfg in "1 1" "2 2" "4 4" "2 4" "4 2"; do
  set -- $cfg
  EDGEDICT_LPW_SUB=$1 EDGEDICT_SK_SUB=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-fp32-run > gpurun_out/r4a/bench_sub$1_$2.json 2> gpurun_out/r4a/bench_sub$1_$2.err
done
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4a/test_all.txt
