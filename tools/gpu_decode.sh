#!/bin/bash
# gpurun: decode tests + decode / stream throughput -> gpurun_out/<tag>   (usage: bash tools/gpu_decode.sh r4j)
TAG=${1:-r4j}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_beam_gpu.py tests/test_stream_gpu.py tests/test_models_gpu.py -x -q 2>&1 | tail -8 > $OUT/pytest.txt
timeout 600 python tools/decode_bench.py > $OUT/decode_bench.txt 2>&1
timeout 600 python tools/stream_bench.py bf16 > $OUT/stream_bench.txt 2>&1
