#!/usr/bin/env python3
"""SURVEY.md §8d config 5 with the U-Net: a synthetic 10,240 x 10,240 x 5 uint8 raster through the device zone_detect
pipeline (512-pixel windows, margin 128 -> 1600 windows), plus the per-batch cost of the device data feed.
Prints one JSON line.  MODEL=segformer runs the same raster through SegFormer-MiT-B2 (5 channels, 19 labels: BASELINE
config 5 as named), BATCH windows per forward (default 32 for the U-Net, 16 for SegFormer); KERNELS=1 adds the per-kernel table
of one SegFormer batch (HIP events inside the library)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flair-1_amd"))
import flair_amd  # noqa: E402
from flair_amd.data_feed import TileFeed, draw_d4  # noqa: E402
from flair_amd.zone_detect import ZoneDetector, tile_grid  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    side = int(os.environ.get("RASTER", "10240"))
    C = 19
    torch.manual_seed(2022)
    which = os.environ.get("MODEL", "unet")
    dt_name = os.environ.get("DTYPE", "bf16")
    if which == "segformer":
        model = flair_amd.SegformerForSemanticSegmentation(num_channels=5, num_labels=C, compute_dtype=dt_name).to(dev).eval()
    else:
        model = flair_amd.create_model("unet", "resnet34", encoder_weights=None, in_channels=5, classes=C, compute_dtype=dt_name).to(dev).eval()
    batch = int(os.environ.get("BATCH", "16" if which == "segformer" else "32"))
    cfg = {"img_pixels_detection": 512, "margin": 128, "output_type": "argmax", "n_classes": C, "batch_size": batch,
           "channels": [1, 2, 3, 4, 5],
           "norma_task": [{"norm_type": "custom", "norm_means": [105.08, 110.87, 101.82, 106.38, 53.26],
                           "norm_stds": [52.17, 45.38, 44, 39.69, 79.3]}]}
    raster = torch.randint(0, 256, (5, side, side), dtype=torch.uint8, device=dev)
    det = ZoneDetector(model, cfg)
    n = len(tile_grid((side, side), 512, 128))
    det.run(raster[:, :2048, :2048].contiguous())  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = det.run(raster)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    host = out.cpu()
    d2h = time.perf_counter() - t1
    # data feed: 32 stored tiles -> batch["img"], batch["msk"]
    img = torch.randint(0, 256, (32, 5, 512, 512), dtype=torch.uint8, device=dev)
    raw = torch.randint(0, 20, (32, 512, 512), dtype=torch.uint8, device=dev)
    feed = TileFeed([1, 2, 3, 4, 5], 13, "custom", cfg["norma_task"][0]["norm_means"], cfg["norma_task"][0]["norm_stds"])
    d4 = draw_d4(32).to(dev)
    for _ in range(3):
        feed(img, raw, d4=d4)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        feed(img, raw, d4=d4)
    e1.record()
    torch.cuda.synchronize()
    feed_ms = e0.elapsed_time(e1) / 20
    feed_bytes = 32 * 512 * 512 * (5 + 1 + 5 * 4 + 1)
    kernels = None
    if which == "segformer" and os.environ.get("KERNELS") == "1":
        import ctypes as C_
        from flair_amd import _lib as L
        x = torch.randn(batch, 5, 512, 512, device=dev)
        model.forward_full(x)
        L.check(L.lib().flair_profile_start(4096))
        model.forward_full(x)
        n_k = L.lib().flair_profile_stop()
        name = C_.create_string_buffer(96)
        ms, fl, by, cnt = C_.c_double(), C_.c_double(), C_.c_double(), C_.c_int64()
        kernels = []
        for i in range(n_k):
            L.lib().flair_profile_kernel(i, name, 96, C_.byref(ms), C_.byref(cnt), C_.byref(fl), C_.byref(by))
            kernels.append({"kernel": name.value.decode(), "ms": round(ms.value, 4), "launches": cnt.value,
                            "tflops": round(fl.value / max(ms.value, 1e-9) / 1e9, 1), "gbps": round(by.value / max(ms.value, 1e-9) / 1e6, 1)})
        kernels.sort(key=lambda r: -r["ms"])
    label = "SegFormer-MiT-B2" if which == "segformer" else "U-Net/ResNet34"
    print(json.dumps({"workload": f"zone_detect {side}x{side}x5 uint8, {label} {dt_name}, 19 classes, argmax, 512/128, batch {batch}",
                      "kernels_one_batch": kernels,
                      "windows": n, "seconds": round(dt, 3), "windows_per_s": round(n / dt, 1),
                      "megapixels_per_s": round(side * side / dt / 1e6, 1), "result_d2h_s": round(d2h, 3),
                      "result_bytes": host.numel() * host.element_size(),
                      "feed_tiles_ms_per_batch32": round(feed_ms, 3), "feed_tiles_GBps": round(feed_bytes / feed_ms / 1e6, 1)}))


if __name__ == "__main__":
    main()
