#!/bin/bash
# rocprofv3 kernel statistics of the training bench for several builds of the library on one box:
#   bash scripts/prof_libs.sh tag[:ENV=val] ...     (tag "base" = the default library) -> gpurun_out/prof_ab/lib_<tag>_kernel_stats.csv
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_ab
mkdir -p "$OUT"
cd /tmp
for spec in "$@"; do
  tag=${spec%%:*}; envs=${spec#*:}
  unset FLAIR_HIP_LIB FLAIR_BWD_FUSE
  lib=${tag%%+*}
  if [ "$lib" != "base" ]; then export FLAIR_HIP_LIB=$REPO/flair-1_amd/flair_amd/libflair_hip_$lib.so; fi
  if [ "$envs" != "$spec" ]; then export $envs; fi
  rm -rf "$OUT/t_$tag"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t_$tag" -- python3 "$REPO/bench.py" --steps 13 --warmup 3 --no-cpu-baseline --train-only --no-alone > "$OUT/lib_$tag.json" 2> "$OUT/lib_$tag.log" || exit 1
  cp "$(find "$OUT/t_$tag" -name "*kernel_stats.csv" | head -1)" "$OUT/lib_${tag}_kernel_stats.csv"
  rm -rf "$OUT/t_$tag"
done
