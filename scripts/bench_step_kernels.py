"""Per-kernel HIP-event times of one training step with the weight-gradient stream off (every kernel alone on the chip):
the table DESIGN.md's "what bounds the step" paragraph is made from.   python scripts/bench_step_kernels.py [filter]"""
import ctypes as C, os, sys
os.environ.setdefault("FLAIR_WGRAD_STREAM", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flair-1_amd"))
import torch, flair_amd
from flair_amd import _lib as L
dev = torch.device("cuda:0")
torch.manual_seed(2022)
m = flair_amd.create_model("unet", "resnet34", encoder_weights=None, in_channels=5, classes=13, compute_dtype="bf16").to(dev).train()
tr = flair_amd.SegTrainer(m, lr=0.02)
x = torch.randn(32, 5, 512, 512, device=dev); y = torch.randint(0, 13, (32, 512, 512), device=dev, dtype=torch.uint8)
for _ in range(3): tr.train_step(x, y)
torch.cuda.synchronize()
L.check(L.lib().flair_profile_start(4096))
tr.train_step(x, y); torch.cuda.synchronize()
n = L.lib().flair_profile_stop()
name = C.create_string_buffer(96); ms, fl, by, cnt = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
rows = []
for i in range(n):
    L.lib().flair_profile_kernel(i, name, 96, C.byref(ms), C.byref(cnt), C.byref(fl), C.byref(by))
    rows.append((name.value.decode(), ms.value, cnt.value, fl.value, by.value))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
tot = sum(r[1] for r in rows)
print(f"sum of kernel times {tot:.3f} ms")
for nm, t, c, f, b in sorted(rows, key=lambda r: -r[1]):
    if flt in nm:
        print(f"{nm:32s} {c:4d} x {t / c * 1e3:8.1f} us = {t:7.3f} ms  {f / t / 1e9 if t else 0:8.1f} TF  {b / t / 1e9 if t else 0:7.2f} TB/s")
