set -e
REPO=$(pwd); export TMPDIR=/tmp; OUT=$REPO/gpurun_out/trace_sf; rm -rf $OUT; mkdir -p $OUT
python3 scripts/bench_segformer.py > $OUT/bench.json
cd /tmp
REPS=2 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $REPO/scripts/bench_segformer.py > $OUT/bench_prof.json 2> $OUT/log.txt
cd $REPO
F=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 scripts/trace_segformer.py $F > $OUT/dump.txt
find $OUT -name "*kernel_trace.csv" -size +20M -delete
cat $OUT/bench.json; tail -25 $OUT/dump.txt
