#!/bin/bash
mkdir -p gpurun_out/r4j
cd /root/repo
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_stream_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/r4j/pytest.txt
timeout 600 python tools/stream_bench.py > gpurun_out/r4j/stream_bench.txt 2>&1
EDGEDICT_STREAM_ENCODER_STEP=0 timeout 600 python tools/stream_bench.py > gpurun_out/r4j/stream_bench_old.txt 2>&1
