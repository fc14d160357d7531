#!/bin/bash
mkdir -p gpurun_out/r4q
cd /root/repo
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r4q/test_all.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4q/bench.json 2> gpurun_out/r4q/bench.err
