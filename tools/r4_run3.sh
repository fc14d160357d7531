#!/bin/bash
mkdir -p gpurun_out/r4c
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_lpw_gpu.py -x -q -k "lpw_forward" 2>&1 | tail -5 > gpurun_out/r4c/test_lpw.txt
timeout 300 python tools/fwd_time.py "BWD=1,SUB=1,POLL=0" "BWD=1,SUB=1,POLL=1" > gpurun_out/r4c/fwd_time.txt 2>&1
rm -f gpurun_out/r4c/lpw_trace.txt
EDGEDICT_LPW_SUB=1 EDGEDICT_LPW_POLL=1 timeout 120 python tools/lpw_trace.py >> gpurun_out/r4c/lpw_trace.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-fp32-run > gpurun_out/r4c/bench.json 2> gpurun_out/r4c/bench.err
