import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/flair-1_amd")
import torch, flair_amd
dev = torch.device("cuda:0")
m = flair_amd.create_model("unet", "resnet34", encoder_weights=None, in_channels=5, classes=13, compute_dtype="bf16").to(dev).train()
tr = flair_amd.SegTrainer(m, lr=0.02)
x = torch.randn(32, 5, 512, 512, device=dev); y = torch.randint(0, 13, (32, 512, 512), device=dev, dtype=torch.uint8)
for _ in range(5): tr.train_step(x, y)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); tr.train_step(x, y); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print("host enqueue ms per step:", [round(a * 1e3, 2) for a, b in ts])
print("step wall ms (isolated, with sync):", [round(b * 1e3, 2) for a, b in ts])
