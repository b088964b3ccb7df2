"""Phase anatomy of the halo-GEMM kernel: per-workgroup s_memtime stamps (flair_debug_buffer) -> cycles spent in the
prologue (halo + first weight tiles), the tap loop, accumulator staging and the tile store.  Diagnostics only: needs the
stamp build (FLAIR_STAMPS=1 python flair-1_amd/build.py --force); the shipped library has no stamps."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flair-1_amd"))
import torch
from flair_amd import _lib as L
from flair_amd import ops

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
dt = torch.bfloat16
B = 32
SHAPES = [("l3 256->256 @32", 32, 256, 0, 0, 256), ("l4 512->512 @16", 16, 512, 0, 0, 512), ("l2 128->128 @64", 64, 128, 0, 0, 128),
          ("d0.c1 768->256", 16, 512, 256, 1, 256), ("l1 64->64 @128", 128, 64, 0, 0, 64)]
L.lib().flair_tune_set(b"FLAIR_HG_DMA", mode)
for name, H, C0, C1, up0, Cout in SHAPES:
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(B, H, H, C0, generator=g).to(dev, dt)
    Hin = 2 * H if up0 else H
    x1 = torch.randn(B, Hin, Hin, C1, generator=g).to(dev, dt) if C1 else None
    w = (torch.randn(Cout, C0 + C1, 3, 3, generator=g) / (9 * (C0 + C1)) ** 0.5).to(dev)
    for _ in range(3):
        ops.conv2d_forward(x0, w, x1=x1, up0=bool(up0), want_stats=True)
    buf = torch.zeros(8192, 8, dtype=torch.int64, device=dev)
    L.check(L.lib().flair_debug_buffer(L.ptr(buf)), "flair_debug_buffer (diagnostic build needed)")
    ops.conv2d_forward(x0, w, x1=x1, up0=bool(up0), want_stats=True)
    torch.cuda.synchronize()
    L.lib().flair_debug_buffer(None)
    s = buf.cpu()
    s = s[s[:, 0] != 0].double()
    n = s.shape[0]
    d = [(s[:, i + 1] - s[:, i]) for i in range(4)]
    tot = s[:, 4] - s[:, 0]
    med = lambda v: float(v.median())
    nch = (C0 + C1) // 64
    rt0, rt1 = s[:, 5], s[:, 6]
    span_us = (float(rt1.max()) - float(rt0.min())) / 100.0
    clk = float(((s[:, 4] - s[:, 0]) / (rt1 - rt0)).median()) * 100.0   # MHz
    starts = ((rt0 - rt0.min()) / 100.0)
    late = int((starts > 0.25 * span_us).sum())
    print(f"    realtime: kernel span {span_us:6.1f} us, median WG lifetime {float((rt1 - rt0).median()) / 100.0:6.1f} us, in-kernel clock {clk:6.0f} MHz, "
          f"WGs starting after 25% of the span: {late}")
    print(f"    staging writes done {med(s[:, 7] - s[:, 2]):6.0f} cyc after the loop; statistics + barrier {med(s[:, 3] - s[:, 7]):6.0f}")
    print(f"{name:20s} {n} WGs  total {med(tot):8.0f} cyc | prologue {med(d[0]):7.0f} | loop {med(d[1]):8.0f} = {med(d[1]) / (9 * nch):6.0f}/tap "
          f"| stage {med(d[2]):6.0f} | store {med(d[3]):6.0f} | span(max end - min start) {float(s[:, 4].max() - s[:, 0].min()):9.0f}", flush=True)
