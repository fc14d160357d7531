#!/bin/bash
# gpurun: a pytest selection N times, full output of failing runs kept -> gpurun_out/r4flake4   (usage: bash tools/gpu_flake.sh N "pytest args")
N=${1:-4}
ARGS=${2:-"tests -m gpu"}
cd /root/repo; mkdir -p gpurun_out/r4flake4; export TMPDIR=/tmp
for i in $(seq 1 $N); do
  timeout 900 python -m pytest $ARGS -x -q > /tmp/fl_$i.txt 2>&1
  tail -n 1 /tmp/fl_$i.txt >> gpurun_out/r4flake4/summary.txt
  if grep -q "failed" /tmp/fl_$i.txt; then tail -n 120 /tmp/fl_$i.txt > gpurun_out/r4flake4/fail_$i.txt; fi
done
