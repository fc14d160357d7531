#!/bin/bash
# gpurun: the profile set of a round -> gpurun_out/<tag>_*   (usage: bash tools/gpu_profiles.sh r4b)
TAG=${1:-r4b}
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
bash profiles/collect.sh $TAG
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 300 python tools/overlap_report.py > gpurun_out/${TAG}_overlap.txt 2>&1
timeout 300 python tools/host_profile.py > gpurun_out/${TAG}_host.txt 2>&1
timeout 120 python tools/lpw_trace.py > gpurun_out/${TAG}_lpw_trace.txt 2>&1
timeout 120 python tools/sk_trace.py > gpurun_out/${TAG}_sk_trace.txt 2>&1
