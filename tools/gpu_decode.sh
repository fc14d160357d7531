#!/bin/bash
mkdir -p gpurun_out/r4j
cd /root/repo
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_beam_gpu.py tests/test_stream_gpu.py tests/test_models_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r4j/pytest.txt
timeout 600 python tools/decode_bench.py > gpurun_out/r4j/decode_bench.txt 2>&1
