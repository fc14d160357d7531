#!/bin/bash
# On the GPU box: the RNN-T GEMM shapes with the 256x256 kernel on/off and the vendor route on/off.
cd ${GRAFT_REPO_ROOT:-.}
for cfg in "1 1" "0 1" "1 0" "0 0"; do
  set -- $cfg
  echo "== EDGEDICT_GEMM_NT256=$1 EDGEDICT_BLASLT=$2"
  EDGEDICT_GEMM_NT256=$1 EDGEDICT_BLASLT=$2 EDGEDICT_BLASLT_SMALL=$2 python tools/gemm_bench.py 2>/dev/null | grep -E "logits|dhid via|L1 input|square"
done
