#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel of one csrc file (hipcc -Rpass-analysis=kernel-resource-usage), one line each.
    python scripts/kernel_regs.py conv_hg.hip [substring]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "flair-1_amd", "csrc", sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:], capture_output=True, text=True)
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line) or re.search(r"remark: .*? Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).split(" [")[0]] = int(m.group(2))
if r.returncode:
    print(r.stderr[-3000:])
for c in rows:
    if pat in c["name"]:
        nm = re.sub(r"flair::\(anonymous namespace\)::|flair::|void ", "", c["name"])
        print(f"{nm[:100]:100s} v{c.get('VGPRs',0):4d} a{c.get('AGPRs',0):4d} s{c.get('TotalSGPRs',0):4d} scratch{c.get('ScratchSize',0):5d} vspill{c.get('VGPRs Spill',0):3d} occ{c.get('Occupancy',0)}")
