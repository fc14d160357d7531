#!/bin/bash
# gpurun -- 'bash tools/gpu_r5d.sh <tag>': encoder-stack tests (dropout inside the stack), then the bench line
tag=${1:-r5d}
cd /root/repo; mkdir -p gpurun_out/$tag; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_encoder_stack_gpu.py tests/test_lpw_gpu.py tests/test_train_step_gpu.py tests/test_e6d2_parity_gpu.py tests/test_models_gpu.py -m gpu -q -p no:cacheprovider > /tmp/t.txt 2>&1; tail -n 70 /tmp/t.txt > gpurun_out/$tag/tests.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-secondary --no-fp32-run --no-own-kernels-run > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
tail -n 6 gpurun_out/$tag/tests.txt; cut -c1-260 gpurun_out/$tag/bench.json
