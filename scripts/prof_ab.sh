#!/bin/bash
# rocprofv3 kernel statistics of the training bench under two settings of one tuning switch (same box, back to back):
#   bash scripts/prof_ab.sh FLAIR_BWD_FUSE 1 0      -> gpurun_out/prof_ab/<KEY>_<value>_kernel_stats.csv
KEY=$1; shift
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_ab
mkdir -p "$OUT"
cd /tmp
for v in "$@"; do
  rm -rf "$OUT/t_$v"
  export $KEY=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t_$v" -- python3 "$REPO/bench.py" --steps 13 --warmup 3 --no-cpu-baseline --train-only --no-alone > "$OUT/${KEY}_${v}.json" 2> "$OUT/${KEY}_${v}.log" || exit 1
  S=$(find "$OUT/t_$v" -name "*kernel_stats.csv" | head -1)
  cp "$S" "$OUT/${KEY}_${v}_kernel_stats.csv"
  rm -rf "$OUT/t_$v"
done
