#!/usr/bin/env python3
"""Per-stream timeline of one training step from a rocprofv3 kernel trace (csv): busy time of each hardware queue, their
union / overlap / idle time, and the largest gaps of the main queue with the kernels around them.

usage: trace_timeline.py <kernel_trace.csv> [step index from the end, default 1]
"""
import csv, sys, collections, re

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# a step starts at the input layout conversion / stem: find the step boundaries by the sgd kernel
sgd = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]
lo, hi = sgd[-1 - back] + 1, sgd[-back] + 1
step = rows[lo:hi]
t0, t1 = step[0]["s"], max(r["e"] for r in step)
print(f"step: {len(step)} kernels, {(t1 - t0) / 1e6:.3f} ms")


def short(n):
    m = re.search(r"(\w+)(?:<|\()", n.replace("flair::", "").replace("(anonymous namespace)::", "").replace("void ", ""))
    return m.group(1) if m else n[:40]


def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs; cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + ce - cs


byq = collections.defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
allu = union([(r["s"], r["e"]) for r in step])
print(f"GPU busy (union) {allu / 1e6:.3f} ms, idle {(t1 - t0 - allu) / 1e6:.3f} ms")
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    u = union([(r["s"], r["e"]) for r in rs])
    print(f"queue {q}: {len(rs)} kernels, busy {u / 1e6:.3f} ms, first {(rs[0]['s'] - t0) / 1e6:.3f} last {(max(r['e'] for r in rs) - t0) / 1e6:.3f}")
    cat = collections.Counter()
    for r in rs:
        cat[short(r["Kernel_Name"])] += r["e"] - r["s"]
    print("   " + ", ".join(f"{k} {v / 1e6:.2f}" for k, v in cat.most_common(12)))
mainq = max(byq.items(), key=lambda kv: len(kv[1]))[0]
rs = byq[mainq]
gaps = []
for a, b in zip(rs, rs[1:]):
    gaps.append((b["s"] - a["e"], short(a["Kernel_Name"]), short(b["Kernel_Name"]), (a["e"] - t0) / 1e6))
tot_gap = sum(g[0] for g in gaps if g[0] > 0)
print(f"main queue gaps: total {tot_gap / 1e6:.3f} ms over {len(gaps)}; > 10 us: {sum(1 for g in gaps if g[0] > 10000)}")
for g in sorted(gaps, reverse=True)[:15]:
    print(f"   {g[0] / 1e3:8.1f} us after {g[1]} before {g[2]} at {g[3]:.3f} ms")
