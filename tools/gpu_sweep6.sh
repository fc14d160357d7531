#!/bin/bash
# gpurun: round-6 sweep of the remaining schedule knobs of the training step -> gpurun_out/r6_sweep.txt
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
F="--steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-own-kernels-run --no-fp32-run --no-secondary"
run() {
  env "$@" timeout 300 python bench.py $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-44s %.3f ms  fwd %.1f us bwd %.1f us  stack fwd %.2f bwd %.2f' % (' '.join(sys.argv[1:]), d['ms_per_step'], r['fwd_kernel_us'], r['kernel_us'], d['kernel_ms']['enc_stack_fwd_T401_L6'], d['kernel_ms']['enc_stack_bwd_T401_L6']))" "$@" >> gpurun_out/r6_sweep.txt
}
rm -f gpurun_out/r6_sweep.txt
run X=default
run EDGEDICT_STACK_TAIL_SPLIT=2
run EDGEDICT_STACK_TAIL_SPLIT=3
run EDGEDICT_STACK_DW_SEG=8
run EDGEDICT_STACK_CHUNK=20
run EDGEDICT_BENCH_PREFETCH=0
run EDGEDICT_GEMM_NT256R=0
run X=default_again
cat gpurun_out/r6_sweep.txt
