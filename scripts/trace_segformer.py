#!/usr/bin/env python3
"""Kernels of the last SegFormer forward in a rocprofv3 kernel trace (csv), in start order, + totals by kernel name."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
first = [i for i, r in enumerate(rows) if "nchw_to_nhwc" in r["Kernel_Name"]]
step = rows[first[-1]:]
t0 = step[0]["s"]
tot = collections.Counter(); cnt = collections.Counter()
for r in step:
    n = r["Kernel_Name"].replace("flair::", "").replace("(anonymous namespace)::", "").replace("void ", "").replace("unsigned short", "bf16")
    n = re.sub(r"\(.*", "", n)
    g = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])); gy = int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"]))
    print(f"{(r['s'] - t0) / 1e3:9.1f} {(r['e'] - r['s']) / 1e3:8.1f} g{g:<6d}x{gy:<4d} {n[:90]}")
    tot[n] += r["e"] - r["s"]; cnt[n] += 1
print("---- span %.3f ms, busy %.3f ms" % ((step[-1]["e"] - t0) / 1e6, sum(tot.values()) / 1e6))
for n, v in tot.most_common():
    print(f"{v / 1e6:8.3f} ms {cnt[n]:4d}  {n[:100]}")
