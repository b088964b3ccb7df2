#!/bin/bash
# Dynamic instruction mix per kernel (one PMC pass): VALU / SALU / LDS / VMEM instructions per wave.
set -e
REPO=$(pwd); export TMPDIR=/tmp; OUT=$REPO/gpurun_out/pmc_insts; rm -rf $OUT; mkdir -p $OUT; cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA --output-format csv -d $OUT/p -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --train-only --no-alone > /dev/null 2> $OUT/log.txt
cd $REPO
F=$(find $OUT/p -name "*counter_collection.csv" | head -1)
python3 - "$F" > $OUT/insts.txt <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "flair" not in k: continue
    k = re.sub(r"\(.*", "", k.replace("flair::", "").replace("(anonymous namespace)::", "").replace("void ", ""))[:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen: seen.add(key); n[k] += 1
print(f"{'kernel':70s} launches waves/launch  VALU/wave SALU/wave LDS/wave VMEMrd/wave VMEMwr/wave MFMA/wave")
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    w = c.get("SQ_WAVES", 0) or 1
    print(f"{k:70s} {n[k]:5d} {w / n[k]:10.0f} {c.get('SQ_INSTS_VALU', 0) / w:9.0f} {c.get('SQ_INSTS_SALU', 0) / w:9.0f} {c.get('SQ_INSTS_LDS', 0) / w:8.0f} {c.get('SQ_INSTS_VMEM_RD', 0) / w:8.1f} {c.get('SQ_INSTS_VMEM_WR', 0) / w:8.1f} {c.get('SQ_INSTS_MFMA', 0) / w:8.0f}")
PY
find $OUT -name "*counter_collection.csv" -delete
cat $OUT/insts.txt
