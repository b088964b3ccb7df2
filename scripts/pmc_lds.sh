set -e
REPO=$(pwd); export TMPDIR=/tmp; OUT=$REPO/gpurun_out/pmc_step; rm -rf $OUT; mkdir -p $OUT; cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/p -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --train-only --no-alone > /dev/null 2> $OUT/log.txt
cd $REPO
F=$(find $OUT/p -name "*counter_collection.csv" | head -1)
python3 - "$F" > $OUT/lds.txt <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "flair" not in k: continue
    k = re.sub(r"\(.*", "", k.replace("flair::", "").replace("(anonymous namespace)::", "").replace("void ", ""))[:66]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen: seen.add(key); n[k] += 1
print(f"{'kernel':66s} launches  kcycles  MFMA-busy%  LDS-active%  conflict/active  VALU/wave")
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    cyc = c.get("GRBM_GUI_ACTIVE", 1) / 8 / n[k]           # per launch, per XCD
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / n[k] / 1024  # per SIMD
    lds = c.get("SQ_LDS_IDX_ACTIVE", 0) / n[k] / 256         # per CU
    cf = c.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 1))
    print(f"{k:66s} {n[k]:5d} {cyc / 1e3:8.1f} {100 * mf / cyc:9.1f} {100 * lds / cyc:11.1f} {cf:14.2f} {c.get('SQ_INSTS_VALU', 0) / max(1, c.get('SQ_WAVES', 1)):10.0f}")
PY
find $OUT -name "*counter_collection.csv" -delete
head -40 $OUT/lds.txt
