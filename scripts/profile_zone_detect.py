import ctypes as C, os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/flair-1_amd")
import torch, flair_amd
from flair_amd import _lib as L
from flair_amd.zone_detect import ZoneDetector
dev = torch.device("cuda:0")
Cc = 19
model = flair_amd.create_model("unet", "resnet34", encoder_weights=None, in_channels=5, classes=Cc, compute_dtype="bf16").to(dev).eval()
cfg = {"img_pixels_detection": 512, "margin": 128, "output_type": "argmax", "n_classes": Cc, "batch_size": 32,
       "channels": [1, 2, 3, 4, 5], "norma_task": [{"norm_type": "scaling"}]}
det = ZoneDetector(model, cfg)
r = torch.randint(0, 255, (5, 2560, 2560), dtype=torch.uint8, device=dev)
for _ in range(2): det.run(r)
torch.cuda.synchronize()
L.check(L.lib().flair_profile_start(4096))
det.run(r); torch.cuda.synchronize()
n = L.lib().flair_profile_stop()
name = C.create_string_buffer(96); ms, fl, by, cnt = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
rows = []
for i in range(n):
    L.lib().flair_profile_kernel(i, name, 96, C.byref(ms), C.byref(cnt), C.byref(fl), C.byref(by))
    rows.append((ms.value, name.value.decode(), cnt.value))
for t, k, c in sorted(rows, reverse=True)[:14]: print(f"{k:30s} {t:8.3f} ms {c:4d}")
print("total", sum(r[0] for r in rows))
