#!/bin/bash
# gpurun: streaming decode tests + throughput -> gpurun_out/<tag>   (usage: bash tools/gpu_stream.sh r4p)
TAG=${1:-r4p}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_stream_gpu.py -x -q 2>&1 | tail -12 > $OUT/pytest_stream.txt
timeout 300 python tools/stream_bench.py bf16 > $OUT/stream_bench.txt 2>&1
timeout 300 python tools/stream_host_profile.py 256 > $OUT/stream_host.txt 2>&1
