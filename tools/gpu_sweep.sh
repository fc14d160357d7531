#!/bin/bash
# gpurun: chunk / margin sweep of the training step at the final kernels -> gpurun_out/r4sweep/sweep.txt
cd /root/repo; mkdir -p gpurun_out/r4sweep; export TMPDIR=/tmp
F="--steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-own-kernels-run --no-fp32-run --no-secondary"
run() {
  env "$@" timeout 300 python bench.py $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-44s %.3f ms  fwd %.1f us x? bwd %.1f us  stack fwd %.2f bwd %.2f' % (' '.join(sys.argv[1:]), d['ms_per_step'], r['fwd_kernel_us'], r['kernel_us'], d['kernel_ms']['enc_stack_fwd_T401_L6'], d['kernel_ms']['enc_stack_bwd_T401_L6']))" "$@" >> gpurun_out/r4sweep/sweep.txt
}
run X=default
run EDGEDICT_STACK_CHUNK=12
run EDGEDICT_STACK_CHUNK=20
run EDGEDICT_STACK_CHUNK=24
run EDGEDICT_STACK_CHUNK=32
run EDGEDICT_LPW_MARGIN=1
run EDGEDICT_LPW_MARGIN=3
run EDGEDICT_LPW_MARGIN_B=1
run EDGEDICT_LPW_MARGIN_B=3
run X=default_again
