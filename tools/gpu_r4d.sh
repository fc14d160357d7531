#!/bin/bash
# gpurun: round-4 decode / fp32 / 3-store BPTT check -> gpurun_out/r4d
OUT=gpurun_out/r4d
mkdir -p $OUT
cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_beam_gpu.py tests/test_stream_gpu.py tests/test_models_gpu.py tests/test_lstm_gpu.py tests/test_lpw_gpu.py tests/test_e6d2_parity_gpu.py -x -q -k "not bigger and not lpw_forward_is" 2>&1 | tail -15 > $OUT/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python tools/decode_bench.py > $OUT/decode_bench.txt 2>&1
cd /tmp
for V in 16 100000; do
  rm -rf /tmp/dprof_s$V
  EDGEDICT_STREAM_STEP_MAX_ROWS=$V timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/dprof_s$V -o t -- python /root/repo/tools/decode_profile.py stream256 > /root/repo/$OUT/stream256_rows$V.log 2>&1
  DB=$(find /tmp/dprof_s$V -name "*results.db" | head -1)
  (cd /root/repo && python profiles/summarize.py $DB $OUT/stream256_rows$V.md "stream256, EDGEDICT_STREAM_STEP_MAX_ROWS=$V")
done
