#!/bin/bash
# Kernel-trace statistics and the two PMC passes (FETCH_SIZE / WRITE_SIZE: one counter per pass, the TCC slots allow no
# more) of the bench command, on the GPU box.  Writes under gpurun_out/; scripts/pmc_traffic.py turns the PMC passes into
# profiles/<round>_pmc_traffic.json, the *_kernel_stats.csv is copied to profiles/ as is.
#   usage: bash scripts/profile_round.sh r2 <commit the build was made from>
set -e
R=${1:-r2}
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_$R
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --steps 13 --warmup 3 --no-cpu-baseline --train-only --no-alone > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --train-only --no-alone > /dev/null 2> "$OUT/fetch.log"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --train-only --no-alone > /dev/null 2> "$OUT/write.log"
cd "$REPO"
F=$(find "$OUT/fetch" -name "*counter_collection.csv" | head -1)
W=$(find "$OUT/write" -name "*counter_collection.csv" | head -1)
python3 scripts/pmc_traffic.py "$F" "$W" "$OUT/${R}_pmc_traffic.json" "${2:-unknown}"
S=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
cp "$S" "$OUT/${R}_bench_bs32_bf16_kernel_stats.csv"
# keep the merge small: the raw traces are tens of MB
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -delete
ls -la "$OUT"
