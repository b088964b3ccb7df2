"""Run ONE conv shape a few times (for rocprofv3 --pmc passes): python scripts/bench_one.py <C> <H> [iters] [mode fwd|dgrad|wgrad]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flair-1_amd"))
import torch
from flair_amd import ops
C, H = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
mode = sys.argv[4] if len(sys.argv) > 4 else "dgrad"
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x = torch.randn(32, H, H, C, generator=g).to(dev, torch.bfloat16)
dy = torch.randn(32, H, H, C, generator=g).to(dev, torch.bfloat16)
w = (torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5).to(dev)
for _ in range(iters):
    if mode == "fwd":
        ops.conv2d_forward(x, w)
    elif mode == "dgrad":
        ops.conv2d_backward(x, w, dy, need_dw=False)
    else:
        ops.conv2d_backward(x, w, dy, need_dx=False)
torch.cuda.synchronize()
print("done")
