#!/usr/bin/env python3
"""Side-by-side per-kernel totals of two rocprofv3 kernel_stats.csv files (ms per step, launches per step).
    python scripts/cmp_stats.py a.csv b.csv [steps_profiled=16]"""
import csv, re, sys
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        n = re.sub(r"flair::\(anonymous namespace\)::|flair::|void |\(.*", "", r["Name"])
        n = n.replace("unsigned short", "bf16")
        d[n] = (int(r["Calls"]), float(r["TotalDurationNs"]))
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 16.0
keys = sorted(set(a) | set(b), key=lambda k: -max(a.get(k, (0, 0))[1], b.get(k, (0, 0))[1]))
ta = tb = 0.0
print(f"{'kernel':70s} {'A n':>5s} {'A ms':>8s} {'A us':>7s} | {'B n':>5s} {'B ms':>8s} {'B us':>7s} | {'B-A ms':>7s}")
for k in keys:
    ca, da = a.get(k, (0, 0.0)); cb, db = b.get(k, (0, 0.0))
    ta += da; tb += db
    if max(da, db) / steps < 2e3: continue
    print(f"{k[:70]:70s} {ca/steps:5.1f} {da/steps/1e6:8.3f} {da/max(ca,1)/1e3:7.1f} | {cb/steps:5.1f} {db/steps/1e6:8.3f} {db/max(cb,1)/1e3:7.1f} | {(db-da)/steps/1e6:7.3f}")
print(f"{'TOTAL':70s} {'':5s} {ta/steps/1e6:8.3f} {'':7s} | {'':5s} {tb/steps/1e6:8.3f} {'':7s} | {(tb-ta)/steps/1e6:7.3f}")
