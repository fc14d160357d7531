#!/bin/bash
# usage: gpurun -- 'bash tools/gpu_quick.sh <tag> <pytest args...>'   -> gpurun_out/<tag>/pytest.txt
tag=$1; shift
mkdir -p gpurun_out/$tag
cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest "$@" 2>&1 | tail -40 > gpurun_out/$tag/pytest.txt
