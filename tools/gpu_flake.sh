#!/bin/bash
# gpurun -- 'bash tools/gpu_flake.sh <tag> <N> [poison]': the whole GPU suite N times in fresh processes (isolation fixture of
# tests/conftest.py on: no internal stream may be left running, no bounded wait may give up), full output of every FAILING
# run kept -> gpurun_out/<tag>/{summary.txt,fail_<i>.txt}; "poison" runs them with EDGEDICT_POISON=1
tag=${1:-flake}; N=${2:-4}; mode=${3:-}
cd /root/repo; mkdir -p gpurun_out/$tag; export TMPDIR=/tmp
[ "$mode" = "poison" ] && export EDGEDICT_POISON=1
for i in $(seq 1 $N); do
  timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -p no:randomly > /tmp/fl_$i.txt 2>&1
  echo "run $i: $(tail -n 1 /tmp/fl_$i.txt)" >> gpurun_out/$tag/summary.txt
  if grep -qE "failed|error" /tmp/fl_$i.txt; then tail -n 200 /tmp/fl_$i.txt > gpurun_out/$tag/fail_$i.txt; fi
done
cat gpurun_out/$tag/summary.txt
