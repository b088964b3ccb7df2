"""Dense product [M = B*H*H][K] x [N][K]^T (a 1x1 convolution through flair_conv2d_forward -> conv_igemm_kernel) over a list of
(H, K, N) shapes, under a tuning switch: kernel time by the library's HIP events, TFLOP/s and algorithmic TB/s.

    python scripts/bench_gemm.py KEY modes "H,K,N H,K,N ..." [B]
"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flair-1_amd"))
import torch
from flair_amd import _lib as L, ops

key = sys.argv[1].encode(); modes = [int(m) for m in sys.argv[2].split(",")]
shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[3].split()]
B = int(sys.argv[4]) if len(sys.argv) > 4 else 32
dev = torch.device("cuda:0"); dt = torch.bfloat16


def kernels():
    n = L.lib().flair_profile_stop()
    name = C.create_string_buffer(96); ms, fl, by, cnt = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
    out = {}
    for i in range(n):
        L.lib().flair_profile_kernel(i, name, 96, C.byref(ms), C.byref(cnt), C.byref(fl), C.byref(by))
        out[name.value.decode()] = ms.value / cnt.value
    return out


for H, K, N in shapes:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, H, K, generator=g).to(dev, dt)
    w = (torch.randn(N, K, 1, 1, generator=g) / K ** 0.5).to(dev)
    M = B * H * H
    gf = 2.0 * M * K * N / 1e9; mb = M * (K + N) * 2 / 1e6
    for m in modes:
        L.lib().flair_tune_set(key, m)
        ts = []
        for r in range(7):
            L.check(L.lib().flair_profile_start(64))
            ops.conv2d_forward(x, w, pad=0, want_stats=False)
            k = kernels()
            if r:
                ts.append(sum(v for n, v in k.items() if "igemm" in n or "gemm" in n))
        ts.sort(); med = ts[len(ts) // 2]
        print(f"M={M:7d} K={K:5d} N={N:5d} {key.decode()}={m}: {med * 1e3:7.1f} us  {gf / med:7.1f} TF  {mb / med / 1e3:5.2f} TB/s  [{', '.join(k)}]", flush=True)
