#!/bin/bash
# gpurun -- 'bash tools/gpu_r5b.sh <tag>': the windowed joint backward - its tests, then the bench line with and without it
tag=${1:-r5b}
cd /root/repo; mkdir -p gpurun_out/$tag; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_joint_windows_gpu.py tests/test_reference_loops_gpu.py tests/test_stream_gpu.py tests/test_train_step_gpu.py tests/test_e6d2_parity_gpu.py -m gpu -q -x -p no:cacheprovider > /tmp/t.txt 2>&1; tail -n 80 /tmp/t.txt > gpurun_out/$tag/tests.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-secondary --no-fp32-run --no-own-kernels-run"
for spec in "0.12,0.4,0.7" "0" "0.2,0.6" "0.1,0.3,0.5,0.75" "0.25"; do
  EDGEDICT_JOINT_WINDOWS="$spec" timeout 300 python bench.py $B > gpurun_out/$tag/bench_$spec.json 2> gpurun_out/$tag/bench_$spec.err
  python - "$spec" gpurun_out/$tag/bench_$spec.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("windows", sys.argv[1], "ms_per_step", round(d["ms_per_step"], 3), "value", round(d["value"], 1), {k: round(v, 3) for k, v in d.get("kernel_ms", {}).items()})
except Exception as e:
    print("windows", sys.argv[1], "FAILED", e)
PY
done | tee gpurun_out/$tag/sweep.txt
tail -n 5 gpurun_out/$tag/tests.txt
