#!/bin/bash
# gpurun: stream encoder step at many rows, bench variance -> gpurun_out/r4f
OUT=gpurun_out/r4f
mkdir -p $OUT
cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_stream_gpu.py -x -q 2>&1 | tail -5 > $OUT/pytest_stream.txt
for V in 16 100000; do
  EDGEDICT_STREAM_STEP_MAX_ROWS=$V timeout 300 python tools/stream_bench.py bf16 > $OUT/stream_bench_rows$V.txt 2>&1
done
F="--no-cpu-baseline --no-loss-delta --no-own-kernels-run --no-fp32-run --no-secondary"
for i in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 $F > $OUT/bench_$i.json 2> $OUT/bench_$i.err
done
timeout 300 python tools/overlap_report.py default > $OUT/overlap_default.txt 2>&1
