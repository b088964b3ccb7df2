#!/usr/bin/env python3
"""Build libflair_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python flair-1_amd/build.py [--force]

Output: flair-1_amd/flair_amd/libflair_hip.so  (git-ignored, travels with gpurun snapshots).
hipcc cross-compiles for gfx950 without a GPU present.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "flair_amd")
OBJ_DIR = os.path.join(HERE, "build")
LIB = os.path.join(OUT_DIR, "libflair_hip.so")
SOURCES = ["conv_igemm.hip", "conv_halo.hip", "conv_hg.hip", "wgrad.hip", "wgrad_halo.hip", "wgrad_hg.hip", "stem.hip", "batchnorm.hip", "misc.hip", "ce_head.hip", "feed.hip", "unet.hip", "capi.hip", "prof.hip", "tune.hip", "metadata_mlp.hip", "segformer_ops.hip", "segformer.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
if os.environ.get("FLAIR_STAMPS") == "1":   # diagnostic build: phase stamps in the halo-GEMM kernel (scripts/stamp_hg.py)
    FLAGS.append("-DFLAIR_HG_STAMPS")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "flair_hip.h"))
    hipcc = _hipcc()
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or not _newer(o, [s] + headers):
            jobs.append([hipcc] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, flush=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or not os.path.exists(LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
