#!/bin/bash
# gpurun: persistent greedy frame loop -> gpurun_out/r4g
OUT=gpurun_out/r4g
mkdir -p $OUT
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_beam_gpu.py tests/test_stream_gpu.py tests/test_models_gpu.py -x -q 2>&1 | tail -15 > $OUT/pytest.txt
timeout 300 python tools/decode_bench.py > $OUT/decode_bench.txt 2>&1
EDGEDICT_DECODE_PERSIST=0 timeout 300 python tools/decode_bench.py > $OUT/decode_bench_persist0.txt 2>&1
timeout 300 python tools/stream_bench.py bf16 > $OUT/stream_bench.txt 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-loss-delta --no-own-kernels-run > $OUT/bench.json 2> $OUT/bench.err
