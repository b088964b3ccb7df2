#!/usr/bin/env python3
"""Every kernel of one training step in start order from a rocprofv3 kernel trace (csv): start offset, duration, queue, grid, name.

usage: trace_dump.py <kernel_trace.csv> [step index from the end, default 1]
"""
import csv, sys, re

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
sgd = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]
lo, hi = sgd[-1 - back] + 1, sgd[-back] + 1
step = rows[lo:hi]
t0 = step[0]["s"]
queues = {q: i for i, q in enumerate(sorted({r["Queue_Id"] for r in step}, key=lambda q: -sum(1 for r in step if r["Queue_Id"] == q)))}
for r in step:
    n = r["Kernel_Name"].replace("flair::", "").replace("(anonymous namespace)::", "").replace("void ", "").replace("unsigned short", "bf16")
    n = re.sub(r"\(.*", "", n)
    print(f"{(r['s'] - t0) / 1e3:9.1f} {(r['e'] - r['s']) / 1e3:8.1f} q{queues[r['Queue_Id']]} g{int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):<6d} {n[:100]}")
