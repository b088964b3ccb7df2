#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; one counter per pass, as the TCC slots demand) of
`bench.py --train-only` into profiles/<round>_pmc_traffic.json: HBM-side bytes per launch for every kernel.

Corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE tallies the
128-byte requests of wide coalesced streams at 64 bytes, so the read side is doubled.  Every kernel here loads
16 bytes per lane, the calibrated pattern.

usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import re
import sys

# rocprofv3 kernel name -> the name bench.py's live event table uses for the same launches
ALIASES = [
    (r"conv3x3_hgd?_kernel<unsigned short, \d+, \d+, 128", "conv3x3_hg_bf16_n128"),
    (r"conv3x3_hgd?_kernel<unsigned short, \d+, \d+, 64", "conv3x3_hg_bf16_n64"),
    (r"conv3x3_hgd?_kernel<float, \d+, \d+, 128", "conv3x3_hg_f32_n128"),
    (r"conv3x3_hgd?_kernel<float, \d+, \d+, 64", "conv3x3_hg_f32_n64"),
    (r"wgrad3x3_big_kernel<unsigned short, 1", "wgrad3x3_big_bf16"),
    (r"wgrad3x3_big_kernel<unsigned short, 2", "wgrad3x3_big_bf16_co64"),
    (r"wgrad_big_reduce_kernel|wgrad_reduce_kernel", "wgrad_reduce"),
]


def short(name):
    for pat, alias in ALIASES:
        if re.search(pat, name):
            return alias
    m = re.search(r"(\w+)(<.*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


def per_kernel(path):
    """launch count and counter sum per kernel; template variants that bench.py reports under one name are merged"""
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if "flair" not in r["Kernel_Name"]:
            continue
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = per_kernel(fetch), per_kernel(write)
    res = {}
    for k in sorted(set(f) | set(w)):
        nf, vf = f.get(k, (0, 0.0))
        nw, vw = w.get(k, (0, 0.0))
        if not nf or not nw:
            continue
        rd = 2.0 * vf / nf * 1024.0
        wr = vw / nw * 1024.0
        res[k] = {"launches_sampled": nf, "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                         "hbm_bytes_per_launch": round(rd + wr)}
    import subprocess
    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:  # noqa: BLE001
        commit = None
    commit = (sys.argv[4] if len(sys.argv) > 4 else commit)
    json.dump({"commit": commit, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --steps 2 --warmup 1 "
                         "--train-only; KiB -> bytes, FETCH_SIZE doubled (gfx950)", "kernels": res}, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_sampled"])[:12]:
        print(f"{k[:60]:60s} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch  x{v['launches_sampled']}")


if __name__ == "__main__":
    main()
