#!/usr/bin/env python3
"""Compile ONE instantiation of the LDS-DMA halo-GEMM kernel (seconds instead of minutes) and print its register use.
    python scripts/one_variant.py "bf16_t, 16, 8, 128, 1, 3, true, XF_APPLY" [-D...]
Leaves the assembly in /tmp/one/ for reading."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "flair-1_amd", "csrc", "conv_hg.hip")
s = open(src).read()
i = s.index('template <typename T>\nstatic int launch_hg_t')
j = s.index('int set_debug_buffer')
inst = sys.argv[1]
s = s[:i] + 'template <typename T>\nstatic int launch_hg_t(int tp, const ConvArgs& a, hipStream_t s) {\n  return launch_hgd_cfg_b<%s>(a, s);\n}\n' % inst + s[j:]
s = s.replace('return dtype == DT_F32 ? launch_hg_t<float>(tp, a, s) : launch_hg_t<bf16_t>(tp, a, s);', 'return launch_hg_t<bf16_t>(tp, a, s);')
os.makedirs("/tmp/one", exist_ok=True)
open("/tmp/one/one_hg.hip", "w").write(s)
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.dirname(src), "-I" + os.path.join(ROOT, "include"),
                    "-c", "one_hg.hip", "-o", "one_hg.o", "-save-temps", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:], cwd="/tmp/one", capture_output=True, text=True)
out = [l.split("remark:")[1].split("[-R")[0].strip() for l in r.stderr.splitlines() if "remark:" in l and re.search(r"VGPRs|Scratch|Occupancy|SGPRs:", l)]
print(" | ".join(out))
if r.returncode:
    print(r.stderr[-4000:])
