#!/bin/bash
# round 4, second GPU pass: data-polling forward kernel - parity, timings, phase trace
mkdir -p gpurun_out/r4b
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_lpw_gpu.py -x -q -k "lpw_forward" 2>&1 | tail -25 > gpurun_out/r4b/test_lpw.txt
timeout 300 python tools/fwd_time.py "BWD=1,SUB=1,POLL=0,SKSUB=1" "BWD=1,SUB=1,POLL=1,SKSUB=1" "BWD=1,SUB=2,POLL=1,SKSUB=1" "BWD=1,SUB=4,POLL=1,SKSUB=1" > gpurun_out/r4b/fwd_time.txt 2>&1
for cfg in "1 0" "1 1" "2 1" "4 1"; do
  set -- $cfg
  EDGEDICT_LPW_SUB=$1 EDGEDICT_LPW_POLL=$2 timeout 120 python tools/lpw_trace.py >> gpurun_out/r4b/lpw_trace.txt 2>&1
done
for cfg in "1 1" "2 1"; do
  set -- $cfg
  EDGEDICT_LPW_SUB=$1 EDGEDICT_LPW_POLL=$2 EDGEDICT_SK_SUB=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-fp32-run > gpurun_out/r4b/bench_sub$1_poll$2.json 2> gpurun_out/r4b/bench_sub$1_poll$2.err
done
