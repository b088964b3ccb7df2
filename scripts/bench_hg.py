"""A/B micro-benchmark of the halo-GEMM conv variants (FLAIR_HG_DMA modes) at the BASELINE layer shapes, all in ONE
process with interleaved rounds (cdna_hip_programming.md §5.4 rule 24).  Kernel time = HIP events around the
launch inside the library (flair_profile_*), so the weight pack of the operator wrapper is not counted.
Also checks every mode's output bit for bit against mode 0 (the accumulation order is the same by design).

    python scripts/bench_hg.py [bf16|f32] [B] [modes, e.g. 0,1,2,3] [rounds]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flair-1_amd"))
import torch
from flair_amd import _lib as L
from flair_amd import ops

dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
modes = [int(m) for m in (sys.argv[3] if len(sys.argv) > 3 else "0,1,2,3").split(",")]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 8
tune_key = (sys.argv[5] if len(sys.argv) > 5 else "FLAIR_HG_DMA").encode()
dev = torch.device("cuda:0")
# name, H(in), C0, C1, up0, Cout, count in the net (fwd + dgrad launches of that shape per step)
SHAPES = [
    ("l3 256->256 @32", 32, 256, 0, 0, 256, 22),
    ("l4 512->512 @16", 16, 512, 0, 0, 512, 10),
    ("l2 128->128 @64", 64, 128, 0, 0, 128, 14),
    ("d0.c1 768->256 @32", 16, 512, 256, 1, 256, 1),
    ("d1.c1 384->128 @64", 32, 256, 128, 1, 128, 1),
    ("l1 64->64 @128", 128, 64, 0, 0, 64, 12),
    ("d2.c1 192->64 @128", 64, 128, 64, 1, 64, 1),
]


def kernel_ms():
    n = L.lib().flair_profile_stop()
    name = C.create_string_buffer(96)
    ms, fl, by, cnt = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
    tot = 0.0
    for i in range(n):
        L.lib().flair_profile_kernel(i, name, 96, C.byref(ms), C.byref(cnt), C.byref(fl), C.byref(by))
        if name.value.decode().startswith("conv3x3_hg"):
            tot += ms.value / cnt.value
    return tot


print(f"dtype {dt}, B={B}, modes {modes}, {rounds} interleaved rounds; per mode: median / min kernel us, TFLOP/s at the median")
wsum = {m: 0.0 for m in modes}
for name, H, C0, C1, up0, Cout, cnt in SHAPES:
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(B, H, H, C0, generator=g).to(dev, dt)
    Hin = 2 * H if up0 else H
    x1 = torch.randn(B, Hin, Hin, C1, generator=g).to(dev, dt) if C1 else None
    w = (torch.randn(Cout, C0 + C1, 3, 3, generator=g) / (9 * (C0 + C1)) ** 0.5).to(dev)
    gf = 2.0 * B * Hin * Hin * Cout * (C0 + C1) * 9 / 1e9
    ref = None
    times = {m: [] for m in modes}
    for r in range(rounds + 1):
        for m in modes:
            L.lib().flair_tune_set(tune_key, m)
            L.check(L.lib().flair_profile_start(64))
            y, _, st = ops.conv2d_forward(x0, w, x1=x1, up0=bool(up0), want_stats=True)
            t = kernel_ms()
            if r == 0:   # warm-up round: correctness
                if ref is None:
                    ref = (y.clone(), st.clone())
                else:
                    same = torch.equal(y.view(torch.int16 if dt == torch.bfloat16 else torch.int32),
                                       ref[0].view(torch.int16 if dt == torch.bfloat16 else torch.int32))
                    ssame = torch.allclose(st, ref[1], rtol=1e-5, atol=1e-3)
                    if not (same and ssame):
                        d = (y.float() - ref[0].float()).abs().max().item()
                        print(f"  !! mode {m} differs from mode {modes[0]} on {name}: max|d|={d:.4g} stats_ok={ssame}", flush=True)
            else:
                times[m].append(t)
    line = f"{name:22s} {gf:7.1f} GF |"
    for m in modes:
        ts = sorted(times[m])
        med = ts[len(ts) // 2]
        wsum[m] += med * cnt
        line += f"  m{m}: {med * 1e3:7.1f} / {ts[0] * 1e3:7.1f} us {gf / med:7.1f} TF |"
    print(line, flush=True)
print("count-weighted ms per step:", {m: round(v, 3) for m, v in wsum.items()})
