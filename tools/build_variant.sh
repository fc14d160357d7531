#!/bin/bash
# Build a variant of libedgedict_hip.so with extra -D flags for ONE source (tuning experiments):
#   tools/build_variant.sh <name> <source.hip> -DED_BCH=5 ...   -> edgedict_amd/csrc/variants/lib_<name>.so
# select it with EDGEDICT_LIB=<path>.  The objects of the default build are reused for everything else.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/edgedict_amd/csrc
NAME=$1; SRC=$2; shift 2
mkdir -p $C/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -x hip -O3 -std=c++17 -fPIC -I $ROOT/include -I $C "$@" -c $C/$SRC -o $C/variants/${NAME}_${SRC%.hip}.o
OBJS=$(ls $C/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/variants/lib_$NAME.so $OBJS $C/variants/${NAME}_${SRC%.hip}.o -ldl
echo $C/variants/lib_$NAME.so
