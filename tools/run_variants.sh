#!/bin/bash
# On the GPU box: bench every tools/variants/*.so (plus the default build) once.
cd "$(dirname "$0")/.."
cp edgedict_amd/csrc/libedgedict_hip.so /tmp/default.so
for v in /tmp/default.so "$@"; do
  cp $v edgedict_amd/csrc/libedgedict_hip.so
  for i in 1 2; do
  timeout 120 python bench.py --no-cpu-baseline --steps 8 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), {k:v for k,v in d['kernel_ms'].items() if 'stack' in k})"
  done
done
cp /tmp/default.so edgedict_amd/csrc/libedgedict_hip.so
