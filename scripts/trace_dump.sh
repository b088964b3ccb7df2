set -e
REPO=$(pwd); export TMPDIR=/tmp; OUT=$REPO/gpurun_out/trace_r3; rm -rf $OUT; mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $REPO/bench.py --steps 4 --warmup 3 --no-cpu-baseline --train-only --no-alone > $OUT/bench.json 2> $OUT/log.txt
cd $REPO
F=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 scripts/trace_dump.py $F 1 > $OUT/dump.txt
python3 scripts/trace_timeline.py $F 1 > $OUT/timeline.txt
find $OUT -name "*kernel_trace.csv" -size +20M -delete
tail -30 $OUT/timeline.txt
