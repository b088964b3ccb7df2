set -e
REPO=$(pwd); export TMPDIR=/tmp; OUT=$REPO/gpurun_out/exch; rm -rf $OUT; mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $REPO/bench.py --steps 8 --warmup 3 --no-cpu-baseline --train-only --no-alone --force-exchange > $OUT/bench.json 2> $OUT/log.txt
cd $REPO
S=$(find $OUT/t -name "*kernel_stats.csv" | head -1)
python3 - "$S" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if 'flair' not in r['Name']:
        print(r['Name'][:90], r['Calls'], round(float(r['TotalDurationNs'])/1e6,3),'ms total', round(float(r['AverageNs'])/1e3,1),'us avg')
PY
find $OUT -name "*kernel_trace.csv" -delete
