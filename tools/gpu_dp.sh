#!/bin/bash
mkdir -p gpurun_out/r4f
cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_dp_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r4f/test_dp.txt
timeout 600 python tools/rccl_footprint.py 0 16 32 64 128 > gpurun_out/r4f/footprint.txt 2>&1
timeout 300 python tools/rccl_footprint.py nccl >> gpurun_out/r4f/footprint.txt 2>&1
