#!/bin/bash
# Build a variant of libedgedict_hip.so with extra -D flags on stack_kernels.hip (experiments):
#   [ED_VARIANT_FILE=gemm] tools/variant.sh NAME -DED_FCH=4 -DED_FWD_OCC=1   ->  tools/variants/NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
file=${ED_VARIANT_FILE:-stack_kernels}
mkdir -p tools/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -x hip -O3 -std=c++17 -fPIC -I include -I edgedict_amd/csrc "$@" \
    -c edgedict_amd/csrc/$file.hip -o /tmp/${file}_$name.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "stack_fwd_kernel\|stack_bwd_kernel" | grep -i "Function Name\|VGPRs:\|AGPRs\|Spill\|Occupancy\|LDS Size" || true
objs=$(ls edgedict_amd/csrc/*.o | grep -v /$file.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/variants/$name.so $objs /tmp/${file}_$name.o
echo built tools/variants/$name.so
