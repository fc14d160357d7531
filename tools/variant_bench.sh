#!/bin/bash
# On the GPU box: bench.py under a list of environment variants, one line each.
#   bash tools/variant_bench.sh "NAME=VAL,NAME2=VAL2" "..." ...
cd ${GRAFT_REPO_ROOT:-.}
for v in "$@"; do
  env $(echo "$v" | tr ',' ' ') timeout 240 python bench.py --no-cpu-baseline --no-loss-delta --steps 12 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-60s %.1f utt/s %.2f ms  %s' % ('$v', d['value'], d['ms_per_step'], {k: round(x,2) for k,x in d['kernel_ms'].items()}))"
done
