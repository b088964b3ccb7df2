"""A/B micro-benchmark of the register-resident weight-gradient kernel (wgrad3x3_big) at the BASELINE layer shapes, in ONE
process with interleaved rounds; the switch is a tuning key (default FLAIR_WG_KMAP).  Kernel time = HIP events around the
launch inside the library (flair_profile_*).  Prints the largest difference between the modes' gradients.

    python scripts/bench_wgrad.py [B] [modes, e.g. 0,1] [rounds] [tune key]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flair-1_amd"))
import torch
from flair_amd import _lib as L
from flair_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
modes = [int(m) for m in (sys.argv[2] if len(sys.argv) > 2 else "0,1").split(",")]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 8
tune_key = (sys.argv[4] if len(sys.argv) > 4 else "FLAIR_WG_KMAP").encode()
dev = torch.device("cuda:0")
dt = torch.bfloat16
# name, H, Cin, Cout, launches per step
SHAPES = [("l3 256->256 @32", 32, 256, 256, 11), ("l4 512->512 @16", 16, 512, 512, 5), ("l2 128->128 @64", 64, 128, 128, 7),
          ("l1 64->64 @128", 128, 64, 64, 6), ("d0.c2 256->256 @32", 32, 256, 256, 1), ("d1.c2 128->128 @64", 64, 128, 128, 1)]


if os.environ.get("SHAPES"):   # "H,Cin,Cout;H,Cin,Cout" (e.g. the small-channel layers: 512,16,16;256,32,32;256,128,32)
    SHAPES = [(f"{c}->{k} @{h}", h, c, k, 1) for h, c, k in (map(int, t.split(",")) for t in os.environ["SHAPES"].split(";"))]


def kernel_ms(prefix):
    n = L.lib().flair_profile_stop()
    name = C.create_string_buffer(96)
    ms, fl, by, cnt = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
    tot = {}
    for i in range(n):
        L.lib().flair_profile_kernel(i, name, 96, C.byref(ms), C.byref(cnt), C.byref(fl), C.byref(by))
        tot[name.value.decode()] = ms.value / cnt.value
    return tot


print(f"B={B}, modes {modes} of {tune_key.decode()}, {rounds} interleaved rounds; median / min kernel us, TFLOP/s at the median")
wsum = {m: 0.0 for m in modes}
for name, H, Cin, Cout, cnt in SHAPES:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, H, Cin, generator=g).to(dev, dt)
    dy = torch.randn(B, H, H, Cout, generator=g).to(dev, dt)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(dev)
    gf = 2.0 * B * H * H * Cout * Cin * 9 / 1e9
    ref = None
    times = {m: [] for m in modes}
    red = {m: [] for m in modes}
    for r in range(rounds + 1):
        for m in modes:
            L.lib().flair_tune_set(tune_key, m)
            L.check(L.lib().flair_profile_start(64))
            _, dw = ops.conv2d_backward(x, w, dy, need_dx=False, need_dw=True)
            t = kernel_ms("wgrad")
            if r == 0:
                if ref is None:
                    ref = dw.clone()
                else:
                    d = (dw - ref).abs().max().item()
                    print(f"  mode {m} vs mode {modes[0]} on {name}: max|d|={d:.3g} (scale {ref.abs().max().item():.3g})", flush=True)
            else:
                times[m].append(sum(v for k, v in t.items() if k.startswith("wgrad3x3")))
                red[m].append(t.get("wgrad_reduce", 0.0))
    line = f"{name:22s} {gf:7.1f} GF |"
    for m in modes:
        ts = sorted(times[m]); med = ts[len(ts) // 2]
        rs = sorted(red[m]); rmed = rs[len(rs) // 2]
        wsum[m] += med * cnt
        line += f"  m{m}: {med * 1e3:7.1f} / {ts[0] * 1e3:7.1f} us {gf / med:7.1f} TF  (+reduce {rmed * 1e3:5.1f}) |"
    print(line, flush=True)
print("count-weighted ms per step:", {m: round(v, 3) for m, v in wsum.items()})
