#!/bin/bash
# On the GPU box: forward/backward stack time of bench.py for several (chunk, delay) settings of the
# weights-stationary path.   bash tools/wsr_sweep.sh "12:1 12:2 6:2 4:2"
cd ${GRAFT_REPO_ROOT:-.}
for cfg in $1; do
  c=${cfg%%:*}; d=${cfg##*:}
  EDGEDICT_STACK_CHUNK=$c EDGEDICT_WSR_DELAY=$d timeout 200 python bench.py --no-cpu-baseline --no-loss-delta --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('chunk $c delay $d: %.1f utt/s %.2f ms/step  fwd %.2f bwd %.2f  host %.1f' % (d['value'], d['ms_per_step'], d['kernel_ms'].get('enc_stack_fwd_T401_L6',0), d['kernel_ms'].get('enc_stack_bwd_T401_L6',0), d['host_enqueue_ms_per_step']))"
done
