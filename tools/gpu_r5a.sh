#!/bin/bash
# gpurun -- 'bash tools/gpu_r5a.sh <tag>': GPU suite with the isolation fixture, once more with EDGEDICT_POISON=1, then the bench line
tag=${1:-r5a}
cd /root/repo; mkdir -p gpurun_out/$tag; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > /tmp/suite.txt 2>&1; tail -n 150 /tmp/suite.txt > gpurun_out/$tag/suite.txt
EDGEDICT_POISON=1 timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > /tmp/poison.txt 2>&1; tail -n 150 /tmp/poison.txt > gpurun_out/$tag/suite_poison.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
tail -n 3 gpurun_out/$tag/suite.txt; tail -n 3 gpurun_out/$tag/suite_poison.txt; cut -c1-300 gpurun_out/$tag/bench.json
