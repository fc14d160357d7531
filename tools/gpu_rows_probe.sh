#!/bin/bash
# gpurun: per-phase step times of both recurrence kernels at 64 / 32 / 16 rows -> gpurun_out/<tag>
TAG=${1:-r4u}
cd /root/repo; mkdir -p gpurun_out/$TAG
for B in 64 32 16; do
  EDGEDICT_TRACE_B=$B python tools/lpw_trace.py > gpurun_out/$TAG/lpw_B$B.txt 2>&1
  EDGEDICT_TRACE_B=$B python tools/sk_trace.py > gpurun_out/$TAG/sk_B$B.txt 2>&1
done
python -m pytest tests/test_lpw_gpu.py -x -q -k "forward" 2>&1 | tail -3 > gpurun_out/$TAG/pytest.txt
