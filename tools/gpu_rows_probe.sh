cd /root/repo; mkdir -p gpurun_out/r4u
for B in 64 32 16; do
  EDGEDICT_TRACE_B=$B python tools/lpw_trace.py > gpurun_out/r4u/lpw_B$B.txt 2>&1
  EDGEDICT_TRACE_B=$B python tools/sk_trace.py > gpurun_out/r4u/sk_B$B.txt 2>&1
done
