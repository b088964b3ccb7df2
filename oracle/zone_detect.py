"""TEST INFRASTRUCTURE — CPU restatement of zone_detect's default pipeline (exact clipping, default-sized tiling):
src/zone_detect/slicing_job.py:19-118 (slice_extent), dataset.py:66-113 (boundless window read + normalisation),
compare.py:20-39,69-82 (softmax, margin crop, convert) and the write loop of main.py:397-421, on an in-memory raster
whose geo-transform is the identity (resolution 1, origin at the bottom-left pixel corner).

PARITY UNPINNED for the slicing job: slice_extent needs rasterio / geopandas / shapely, none importable here, and the
reference ships no fixture of a slicing job.  The tile-level arithmetic (normalisation, convert) is pinned — see
oracle/data_feed.py.
"""
import numpy as np
import torch

from . import data_feed


def slice_extent_np(width: int, height: int, patch_size: int, margin: int, stride: int = 0):
    """Rows of the slicing job in its own order, geographic frame (y up): window = [x, x+P] x [y, y+P]; kept box =
    (left, right, bottom, top)."""
    min_x, min_y, max_x, max_y = 0.0, 0.0, float(width), float(height)
    step = float(stride) if stride else float(patch_size - 2 * margin)
    rows, seen = [], set()
    for x in np.arange(min_x - margin, max_x + margin, step):
        if x + patch_size > max_x + margin:
            x = max_x + margin - patch_size
        for y in np.arange(min_y - margin, max_y + margin, step):
            if y + patch_size > max_y + margin:
                y = max_y + margin - patch_size
            left, right = x + margin, min(x + patch_size - margin, max_x)
            bottom, top = y + margin, min(y + patch_size - margin, max_y)
            key = (round(left, 6), round(bottom, 6), round(right, 6), round(top, 6))
            if key not in seen:
                seen.add(key)
                rows.append({"x": x, "y": y, "left": left, "right": right, "bottom": bottom, "top": top})
    return rows


def read_window_np(raster: np.ndarray, bands, x: float, y: float, patch: int) -> np.ndarray:
    """rasterio read(indexes=bands, window=from_bounds(...), boundless=True): pixels outside the raster are 0."""
    H, W = raster.shape[1:]
    c0, r0 = int(round(x)), H - int(round(y)) - patch  # image row of the window's top edge
    out = np.zeros((len(bands), patch, patch), dtype=raster.dtype)
    rs, re = max(r0, 0), min(r0 + patch, H)
    cs, ce = max(c0, 0), min(c0 + patch, W)
    if re > rs and ce > cs:
        out[:, rs - r0:re - r0, cs - c0:ce - c0] = raster[[b - 1 for b in bands], rs:re, cs:ce]
    return out


def detect_raster_np(model, raster: np.ndarray, config: dict, gap_out: np.ndarray = None) -> np.ndarray:
    """gap_out (H, W), optional: receives the top-2 softmax probability gap of the window that owns each pixel (same
    sequential write rule as the result) — what oracle/parity.py needs to tell a tie flip from a wrong mask."""
    P, m = config["img_pixels_detection"], config["margin"]
    norma = config["norma_task"][0]
    H, W = raster.shape[1:]
    mode = config["output_type"]
    out = np.zeros((2, H, W), np.float32) if mode == "argmax" else np.zeros((config["n_classes"], H, W), np.uint8)
    model.eval()
    for row in slice_extent_np(W, H, P, m):
        img = read_window_np(raster, config["channels"], row["x"], row["y"], P)
        x = data_feed.norm_np(img, norma["norm_type"], norma.get("norm_means", []), norma.get("norm_stds", []))
        with torch.no_grad():
            logits = model(torch.from_numpy(x)[None])[0].numpy()
        pred = data_feed.detect_tile_np(logits, m, mode)
        r0, r1 = H - int(round(row["top"])), H - int(round(row["bottom"]))
        c0, c1 = int(round(row["left"])), int(round(row["right"]))
        out[:, r0:r1, c0:c1] = pred[:, :r1 - r0, :c1 - c0]
        if gap_out is not None:
            from . import parity
            gp = parity.top2_gap(logits, axis=0)[m:P - m, m:P - m]
            gap_out[r0:r1, c0:c1] = gp[:r1 - r0, :c1 - c0]
    return out
