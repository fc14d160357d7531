#!/bin/bash
OUT=gpurun_out/r4h
mkdir -p $OUT
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_beam_gpu.py tests/test_stream_gpu.py tests/test_models_gpu.py -x -q 2>&1 | tail -8 > $OUT/pytest.txt
timeout 300 python tools/decode_bench.py > $OUT/decode_bench.txt 2>&1
timeout 300 python tools/stream_bench.py bf16 > $OUT/stream_bench.txt 2>&1
cd /tmp
rm -rf /tmp/dprof_g
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/dprof_g -o t -- python /root/repo/tools/decode_profile.py greedy_bf16 > /root/repo/$OUT/greedy_bf16.log 2>&1
DB=$(find /tmp/dprof_g -name "*results.db" | head -1)
(python profiles/summarize.py $DB $OUT/greedy_bf16.md "greedy bf16, 5 passes")
