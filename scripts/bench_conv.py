"""A/B micro-benchmark of one 3x3 convolution shape (forward operator) under a tuning switch, interleaved in one process.
Kernel time = HIP events around the launches inside the library (all profiled kernels of the call are summed).

    python scripts/bench_conv.py H Cin Cout KEY modes [B] [rounds] [lazy]
"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flair-1_amd"))
import torch
from flair_amd import _lib as L, ops

H, Cin, Cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
key = sys.argv[4].encode(); modes = [int(m) for m in sys.argv[5].split(",")]
B = int(sys.argv[6]) if len(sys.argv) > 6 else 32
rounds = int(sys.argv[7]) if len(sys.argv) > 7 else 8
dev = torch.device("cuda:0"); dt = torch.bfloat16
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, H, Cin, generator=g).to(dev, dt)
w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(dev)
gf = 2.0 * B * H * H * Cout * Cin * 9 / 1e9
mb = B * H * H * (Cin + Cout) * 2 / 1e6


def kernels():
    n = L.lib().flair_profile_stop()
    name = C.create_string_buffer(96); ms, fl, by, cnt = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
    out = {}
    for i in range(n):
        L.lib().flair_profile_kernel(i, name, 96, C.byref(ms), C.byref(cnt), C.byref(fl), C.byref(by))
        out[name.value.decode()] = ms.value / cnt.value
    return out


ref = None; times = {m: [] for m in modes}; names = {}
for r in range(rounds + 1):
    for m in modes:
        L.lib().flair_tune_set(key, m)
        L.check(L.lib().flair_profile_start(64))
        y, _, st = ops.conv2d_forward(x, w, want_stats=os.environ.get("STATS", "1") != "0")
        k = kernels()
        if r == 0:
            names[m] = list(k)
            if ref is None: ref = y.clone()
            else: print(f"mode {m} vs {modes[0]}: max|d| = {(y.float() - ref.float()).abs().max().item():.4g}")
        else:
            times[m].append(sum(k.values()))
print(f"{Cin}->{Cout} @{H}^2 B={B}: {gf:.1f} GF, {mb:.0f} MB algorithmic")
for m in modes:
    ts = sorted(times[m]); med = ts[len(ts) // 2]
    print(f"  {key.decode()}={m}: {med * 1e3:7.1f} us (min {ts[0] * 1e3:.1f})  {gf / med:7.1f} TF  {mb / med / 1e3:6.2f} TB/s   {names[m]}")
