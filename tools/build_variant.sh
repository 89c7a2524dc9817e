#!/bin/bash
# Developer tool: A/B builds of one translation unit.  tools/build_variant.sh <tag> <file.hip> [-DNAME=VALUE ...]
# -> tools/variants/libqingdai_hip_<tag>.so (git-ignored; select it with QD_LIB_PATH)
set -e
TAG=$1; SRC=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/qingdai_amd/csrc
mkdir -p $ROOT/tools/variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-result -Wno-unused-value"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $C/$SRC -o $ROOT/tools/variants/${SRC%.hip}_$TAG.o
OBJS=""
for f in $C/*.o; do b=$(basename $f); if [ "$b" != "${SRC%.hip}.o" ]; then OBJS="$OBJS $f"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $ROOT/tools/variants/${SRC%.hip}_$TAG.o -o $ROOT/tools/variants/libqingdai_hip_$TAG.so -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built tools/variants/libqingdai_hip_$TAG.so
