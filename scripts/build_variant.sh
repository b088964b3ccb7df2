#!/bin/bash
# Build a variant of the library that differs in ONE source file's compile flags (same-box A/B of two builds, FLAIR_HIP_LIB):
#   bash scripts/build_variant.sh <tag> <file.hip> -DFOO=1 ...   -> flair-1_amd/flair_amd/libflair_hip_<tag>.so
set -e
TAG=$1; SRC=$2; shift 2
cd "$(dirname "$0")/../flair-1_amd"
python build.py > /dev/null
mkdir -p build/var_$TAG
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c csrc/$SRC -o build/var_$TAG/${SRC%.hip}.o
OBJS=""
for f in build/*.o; do
  if [ "$(basename $f)" == "${SRC%.hip}.o" ]; then OBJS="$OBJS build/var_$TAG/${SRC%.hip}.o"; else OBJS="$OBJS $f"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o flair_amd/libflair_hip_$TAG.so $OBJS
echo flair-1_amd/flair_amd/libflair_hip_$TAG.so
