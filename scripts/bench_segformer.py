#!/usr/bin/env python3
"""SegFormer-MiT-B2 forward alone (BATCH windows of 512x512x5, 19 labels): ms per forward by HIP events, FULL=1 adds the x4 upsample.

Under `rocprofv3 --kernel-trace` the kernels of the LAST forward are listed by scripts/trace_segformer.sh.
"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flair-1_amd")]
import torch
import flair_amd

B = int(os.environ.get("BATCH", 32)); dt = os.environ.get("DTYPE", "bf16"); full = os.environ.get("FULL", "1") == "1"
reps = int(os.environ.get("REPS", 20))
dev = torch.device("cuda:0")
m = flair_amd.SegformerForSemanticSegmentation(num_channels=5, num_labels=19, compute_dtype=dt).to(dev)
x = torch.randn(B, 5, 512, 512, device=dev)
f = m.forward_full if full else (lambda t: m(t).logits)
for _ in range(3):
    f(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    f(x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(json.dumps({"batch": B, "dtype": dt, "full": full, "ms_per_forward": round(ms, 3), "windows_per_s": round(B / ms * 1e3, 1)}))
