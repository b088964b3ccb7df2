"""Per-layer micro-benchmark of the conv kernels at the BASELINE shapes (B=32, 512x512 input): forward,
data gradient and weight gradient of every distinct conv shape of Unet(resnet34), timed with HIP events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flair-1_amd"))
import torch
from flair_amd import ops

dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
# name, H(in), C0, C1, up0, Cout, R, stride, pad, count in net
SHAPES = [
    ("stem 7x7 s2 8->64", 512, 8, 0, 0, 64, 7, 2, 3, 1),
    ("l1 64->64", 128, 64, 0, 0, 64, 3, 1, 1, 6),
    ("l2.0 64->128 s2", 128, 64, 0, 0, 128, 3, 2, 1, 1),
    ("l2 128->128", 64, 128, 0, 0, 128, 3, 1, 1, 7),
    ("l2 ds 1x1 s2", 128, 64, 0, 0, 128, 1, 2, 0, 1),
    ("l3.0 128->256 s2", 64, 128, 0, 0, 256, 3, 2, 1, 1),
    ("l3 256->256", 32, 256, 0, 0, 256, 3, 1, 1, 11),
    ("l4.0 256->512 s2", 32, 256, 0, 0, 512, 3, 2, 1, 1),
    ("l4 512->512", 16, 512, 0, 0, 512, 3, 1, 1, 5),
    ("d0.c1 768->256", 16, 512, 256, 1, 256, 3, 1, 1, 1),
    ("d1.c1 384->128", 32, 256, 128, 1, 128, 3, 1, 1, 1),
    ("d2.c1 192->64", 64, 128, 64, 1, 64, 3, 1, 1, 1),
    ("d2.c2 64->64@128", 128, 64, 0, 0, 64, 3, 1, 1, 1),
    ("d3.c1 128->32", 128, 64, 64, 1, 32, 3, 1, 1, 1),
    ("d3.c2 32->32", 256, 32, 0, 0, 32, 3, 1, 1, 1),
    ("d4.c1 32->16", 256, 32, 0, 1, 16, 3, 1, 1, 1),
    ("d4.c2 16->16", 512, 16, 0, 0, 16, 3, 1, 1, 1),
]
only = sys.argv[3] if len(sys.argv) > 3 else None


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
print(f"{'layer':22s} {'GFLOP':>8s} | {'fwd ms':>8s} {'TF/s':>7s} | {'dgrad ms':>8s} {'TF/s':>7s} | {'wgrad ms':>8s} {'TF/s':>7s}")
for name, H, C0, C1, up0, Cout, R, stride, pad, cnt in SHAPES:
    if only and only not in name:
        continue
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(B, H, H, C0, generator=g).to(dev, dt)
    Hin = 2 * H if up0 else H
    x1 = torch.randn(B, Hin, Hin, C1, generator=g).to(dev, dt) if C1 else None
    w = (torch.randn(Cout, C0 + C1, R, R, generator=g) / (R * R * (C0 + C1)) ** 0.5).to(dev)
    Ho = (Hin + 2 * pad - R) // stride + 1
    gf = 2.0 * B * Ho * Ho * Cout * (C0 + C1) * R * R / 1e9
    t_f = timeit(lambda: ops.conv2d_forward(x0, w, stride=stride, pad=pad, x1=x1, up0=bool(up0), want_stats=True))
    line = f"{name:22s} {gf:8.1f} | {t_f:8.3f} {gf / t_f:7.1f} |"
    tot["fwd"] += t_f * cnt
    if not C1 and not up0:
        dy = torch.randn(B, Ho, Ho, Cout, generator=g).to(dev, dt)
        t_d = timeit(lambda: ops.conv2d_backward(x0, w, dy, stride=stride, pad=pad, need_dw=False))
        t_w = timeit(lambda: ops.conv2d_backward(x0, w, dy, stride=stride, pad=pad, need_dx=False))
        tot["dgrad"] += t_d * cnt; tot["wgrad"] += t_w * cnt
        line += f" {t_d:8.3f} {gf / t_d:7.1f} | {t_w:8.3f} {gf / t_w:7.1f}"
    print(line, flush=True)
print("weighted totals (ms per step, incl. weight pack per call):", {k: round(v, 2) for k, v in tot.items()})
