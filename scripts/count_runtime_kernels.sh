#!/bin/bash
# HIP-runtime blit kernels (event signalling, small copies) per configuration:  bash scripts/count_runtime_kernels.sh "<bench args>" VAR=value ...
set -e
ARGS="$1"; shift
REPO=$(pwd); export TMPDIR=/tmp; OUT=$REPO/gpurun_out/cnt; rm -rf $OUT; mkdir -p $OUT; cd /tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $REPO/bench.py --steps 8 --warmup 3 --no-cpu-baseline --train-only --no-alone $ARGS > $OUT/bench.json 2> $OUT/log.txt
S=$(find $OUT/t -name "*kernel_stats.csv" | head -1)
echo "args: $ARGS env: $@"; grep -i "rocclr" $S | cut -d, -f1-4
python3 -c "import json; print('ms_per_step', json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])['ms_per_step'])"
find $OUT -name "*kernel_trace.csv" -delete
