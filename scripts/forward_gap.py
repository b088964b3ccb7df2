"""Wall time of the training-mode forward vs the sum of its kernels' HIP-event times (inter-kernel gaps)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flair-1_amd"))
import torch, flair_amd
from flair_amd import _lib as L
dev = torch.device("cuda:0")
torch.manual_seed(2022)
m = flair_amd.create_model("unet", "resnet34", encoder_weights=None, in_channels=5, classes=13, compute_dtype="bf16").to(dev).train()
x = torch.randn(32, 5, 512, 512, device=dev)
for _ in range(3):
    m._c_forward(x, training=True, want_logits=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    m._c_forward(x, training=True, want_logits=False)
e1.record(); torch.cuda.synchronize()
wall = e0.elapsed_time(e1) / 20
L.check(L.lib().flair_profile_start(4096))
m._c_forward(x, training=True, want_logits=False)
torch.cuda.synchronize()
n = L.lib().flair_profile_stop()
name = C.create_string_buffer(96); ms, fl, by, cnt = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
tot, launches = 0.0, 0
for i in range(n):
    L.lib().flair_profile_kernel(i, name, 96, C.byref(ms), C.byref(cnt), C.byref(fl), C.byref(by))
    tot += ms.value; launches += cnt.value
print(f"forward wall {wall:.3f} ms, kernel sum {tot:.3f} ms over {launches} profiled launches, gap {wall - tot:.3f} ms")
