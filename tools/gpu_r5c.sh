#!/bin/bash
# gpurun -- 'bash tools/gpu_r5c.sh <tag>': streaming tests, then the chunk step at ring depths 3 / 4 / 6 / 8 of enc_lstm_tile
tag=${1:-r5c}
cd /root/repo; mkdir -p gpurun_out/$tag; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_stream_gpu.py tests/test_reference_loops_gpu.py -m gpu -q -p no:cacheprovider > /tmp/t.txt 2>&1; tail -n 60 /tmp/t.txt > gpurun_out/$tag/tests.txt
for ns in 3 4 6 8; do
  echo "ring stages $ns: $(EDGEDICT_ENC_TILE_NS=$ns timeout 200 python tools/stream_step_time.py 64 256 1024 2>&1 | tail -n 3 | tr '\n' ' ')"
done | tee gpurun_out/$tag/ring_sweep.txt
tail -n 4 gpurun_out/$tag/tests.txt
