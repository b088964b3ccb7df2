"""Device-side head of zone_detect's per-batch inference.

The reference runs the model, takes ``softmax`` over classes, copies the FULL probability tensor to the host
(19 MiB per 512x512 tile at 19 classes) and then, per tile on the CPU, crops the margin and calls ``convert``
(src/zone_detect/compare.py:20-39,71-76; src/zone_detect/dataset.py:11-34).  ``inference`` below keeps the argument
list of the reference function; with ``fused=True`` (default) it returns the cropped, converted tiles
(2 x float32 per kept pixel for 'argmax', C bytes for 'class_prob') so that only those cross PCIe.
Stitching modes that need the uncropped probabilities ('average', 'average_weights', 'max') and the raster
windowing are outside this row.
"""
from __future__ import annotations

import torch

from . import _lib as L

OUTPUT_TYPES = {"argmax": 0, "class_prob": 1}


def detect_convert(logits: torch.Tensor, margin: int, output_type: str) -> torch.Tensor:
    """softmax(dim=1) -> [:, m:S-m, m:S-m] -> convert(., output_type) for a batch of square tiles.
    'argmax' -> float32 (B, 2, K, K); 'class_prob' -> uint8 (B, C, K, K)."""
    if output_type not in OUTPUT_TYPES:
        raise ValueError("The output type has not been interpreted.")
    if logits.dim() != 4 or logits.shape[2] != logits.shape[3]:
        raise ValueError("logits must be (B, C, S, S)")
    logits = logits.detach().float().contiguous()
    B, C, S, _ = logits.shape
    K = S - 2 * int(margin)
    if K < 1:
        raise ValueError("margin leaves no pixel")
    if output_type == "argmax":
        out = torch.empty(B, 2, K, K, dtype=torch.float32, device=logits.device)
    else:
        out = torch.empty(B, C, K, K, dtype=torch.uint8, device=logits.device)
    L.check(L.lib().flair_detect_convert(L.ptr(logits), B, C, S, int(margin), OUTPUT_TYPES[output_type], L.ptr(out),
                                         L.stream()), "flair_detect_convert")
    return out


def inference(device, model, use_gpu: bool, config: dict, samples: dict, fused: bool = True):
    """compare.py:20-39.  Returns (predictions, indices) as numpy arrays like the reference; with ``fused`` the
    predictions are already margin-cropped and converted to ``config['output_type']``."""
    if not use_gpu or torch.device(device).type != "cuda":
        raise RuntimeError("flair_amd.zone_detect.inference runs on a HIP device only")
    imgs = samples["image"].to(device, non_blocking=True)
    with torch.no_grad():
        logits = model(imgs)
        if config.get("model_framework", {}).get("model_provider") == "HuggingFace":
            logits = logits.logits
        if fused:
            predictions = detect_convert(logits, config["margin"], config["output_type"])
        else:
            predictions = torch.softmax(logits, dim=1)
    indices = samples["index"].cpu().numpy()
    return predictions.cpu().numpy(), indices
