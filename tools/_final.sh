#!/bin/bash
# one gpurun call: whole GPU suite (plain), profile set r6c, overlap report, sk/lpw traces, default bench line
cd $GRAFT_REPO_ROOT
S=$(date +%s)
python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r6c_suite.txt
echo "suite wall $(( $(date +%s) - S )) s" >> gpurun_out/r6c_suite.txt
bash profiles/collect.sh r6c > gpurun_out/r6c_collect.log 2>&1
python tools/overlap_report.py default > gpurun_out/r6c_overlap.txt 2>&1
python tools/overlap_report.py "serial front-end" >> gpurun_out/r6c_overlap.txt 2>&1
python tools/overlap_report.py nodefer >> gpurun_out/r6c_overlap.txt 2>&1
python tools/sk_trace.py > gpurun_out/r6c_sk_trace.txt 2>&1
python tools/lpw_trace.py > gpurun_out/r6c_lpw_trace.txt 2>&1
S=$(date +%s)
python bench.py > gpurun_out/r6c_bench.json 2> gpurun_out/r6c_bench.err
echo "bench wall $(( $(date +%s) - S )) s" >> gpurun_out/r6c_suite.txt
cat gpurun_out/r6c_suite.txt
