"""Per-kernel table of one eval-mode forward + argmax (SegTrainer.predict) at the bench configuration."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flair-1_amd"))
import torch, flair_amd
from flair_amd import _lib as L
dev = torch.device("cuda:0")
torch.manual_seed(2022)
m = flair_amd.create_model("unet", "resnet34", encoder_weights=None, in_channels=5, classes=13, compute_dtype="bf16").to(dev).eval()
tr = flair_amd.SegTrainer(m, lr=0.0)
x = torch.randn(32, 5, 512, 512, device=dev)
for _ in range(3):
    tr.predict(x)
torch.cuda.synchronize()
L.check(L.lib().flair_profile_start(4096))
tr.predict(x)
torch.cuda.synchronize()
n = L.lib().flair_profile_stop()
name = C.create_string_buffer(96); ms, fl, by, cnt = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
rows = []
for i in range(n):
    L.lib().flair_profile_kernel(i, name, 96, C.byref(ms), C.byref(cnt), C.byref(fl), C.byref(by))
    rows.append((ms.value, name.value.decode(), cnt.value, fl.value, by.value))
tot = sum(r[0] for r in rows)
for t, k, c, f, b in sorted(rows, reverse=True):
    print(f"{k:32s} {t:7.3f} ms {c:4d} launches {f / t / 1e9 if t else 0:8.1f} TFLOP/s {b / t / 1e6 if t else 0:8.1f} GB/s")
print("total", round(tot, 3), "ms")
