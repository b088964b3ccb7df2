#!/bin/bash
# PMC counters of one micro-benchmark command, averaged per kernel:  bash scripts/pmc_kernel.sh "<counters>" <python script> [args...]
set -e
CNT="$1"; shift
REPO=$(pwd); export TMPDIR=/tmp; OUT=$REPO/gpurun_out/pmc_kernel; rm -rf $OUT; mkdir -p $OUT; cd /tmp
rocprofv3 --pmc $CNT --output-format csv -d $OUT/p -- python3 $REPO/"$@" > $OUT/out.txt 2> $OUT/log.txt
cd $REPO
F=$(find $OUT/p -name "*counter_collection.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "flair" not in k: continue
    k = re.sub(r"\(.*", "", k.replace("flair::", "").replace("(anonymous namespace)::", "").replace("void ", ""))[:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen: seen.add(key); n[k] += 1
for k, c in agg.items():
    print(k, n[k], {cn: round(v / n[k]) for cn, v in c.items()})
PY
find $OUT -name "*counter_collection.csv" -delete
