#!/bin/bash
# one gpurun call on the final build of a round: whole GPU suite (plain, then with EDGEDICT_POISON=1), profile set <tag>
# (kernel stats + PMC passes), overlap report, sk / lpw traces, the default bench line.   bash tools/final_collect.sh r6d
cd $GRAFT_REPO_ROOT
TAG=${1:-r6d}
S=$(date +%s)
python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > gpurun_out/${TAG}_suite.txt
echo "suite wall $(( $(date +%s) - S )) s" >> gpurun_out/${TAG}_suite.txt
S=$(date +%s)
EDGEDICT_POISON=1 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 >> gpurun_out/${TAG}_suite.txt
echo "suite wall (EDGEDICT_POISON=1) $(( $(date +%s) - S )) s" >> gpurun_out/${TAG}_suite.txt
bash profiles/collect.sh ${TAG} > gpurun_out/${TAG}_collect.log 2>&1
python tools/overlap_report.py default > gpurun_out/${TAG}_overlap.txt 2>&1
python tools/overlap_report.py "serial front-end" >> gpurun_out/${TAG}_overlap.txt 2>&1
python tools/overlap_report.py nodefer >> gpurun_out/${TAG}_overlap.txt 2>&1
python tools/sk_trace.py > gpurun_out/${TAG}_sk_trace.txt 2>&1
python tools/lpw_trace.py > gpurun_out/${TAG}_lpw_trace.txt 2>&1
S=$(date +%s)
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench wall $(( $(date +%s) - S )) s" >> gpurun_out/${TAG}_suite.txt
cat gpurun_out/${TAG}_suite.txt
