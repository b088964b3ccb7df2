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

OUTPUT_TYPES = {"argmax": 0, "class_prob": 1}   # the values config["output_type"] may take (compare.py:69-82)
_MODE_PROBS = 2                                   # private: every fp32 softmax probability, no convert (compare.py:35)


def detect_convert(logits: torch.Tensor, margin: int, output_type: str, _probs: bool = False) -> torch.Tensor:
    """softmax(dim=1) -> [:, m:S-m, m:S-m] -> convert(., output_type) for a batch of square tiles.
    'argmax' -> float32 (B, 2, K, K); 'class_prob' -> uint8 (B, C, K, K)."""
    if not _probs and output_type not in OUTPUT_TYPES:
        raise ValueError("The output type has not been interpreted.")
    if logits.dim() != 4 or logits.shape[2] != logits.shape[3]:
        raise ValueError("logits must be (B, C, S, S)")
    logits = logits.detach().float().contiguous()
    B, C, S, _ = logits.shape
    K = S - 2 * int(margin)
    if K < 1:
        raise ValueError("margin leaves no pixel")
    if _probs:
        out = torch.empty(B, C, K, K, dtype=torch.float32, device=logits.device)
    elif output_type == "argmax":
        out = torch.empty(B, 2, K, K, dtype=torch.float32, device=logits.device)
    else:
        out = torch.empty(B, C, K, K, dtype=torch.uint8, device=logits.device)
    L.check(L.lib().flair_detect_convert(L.ptr(logits), B, C, S, int(margin), _MODE_PROBS if _probs else OUTPUT_TYPES[output_type], L.ptr(out),
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
        else:   # the reference's own return value: every probability, uncropped (same HIP kernel, no ATen op on the path)
            predictions = detect_convert(logits, 0, "", _probs=True)
    indices = samples["index"].cpu().numpy()
    return predictions.cpu().numpy(), indices


# ------------------------------------------------------------------------------------------------------------------
# Default pipeline (exact clipping, default-sized tiling) over one raster resident in HBM — main.py:386-428.

def get_stride(config: dict) -> list:
    """src/zone_detect/test/tiles.py:4-15."""
    img_size = config["img_pixels_detection"]
    if not config.get("overlap_strat"):
        return [int(img_size - 2 * config["margin"])]
    return [int(i * img_size) for i in config["strategies"]["tiling"]["stride_range"]]


def _axis_origins(extent: int, patch: int, margin: int, step: int) -> list:
    """Window origins along one axis as slice_extent lays them out (slicing_job.py:52-67): start one margin before the
    raster, advance by ``step``, pull a window that would end past extent + margin back to end exactly there;
    repeats are dropped (slicing_job.py:86-95)."""
    out = []
    for o in range(-margin, extent + margin, step):
        if o + patch > extent + margin:
            o = extent + margin - patch
        if o not in out:
            out.append(o)
    return out


def tile_grid(img_size, patch_size: int, margin: int, stride: int | None = None):
    """Pixel-space slicing job.  img_size = (width, height).  Returns an (n, 6) int32 array of
    {x0, y0, wx0, wx1, wy0, wy1} in the job's order (columns outer, rows inner, rows counted from the BOTTOM of the
    raster like the reference's geographic Y axis): window origin (top-left, image coordinates) and the output
    rectangle the window owns once every later window of the job has been written over it."""
    import numpy as np
    W, H = int(img_size[0]), int(img_size[1])
    K = patch_size - 2 * margin
    step = int(stride) if stride else K
    if K < 1 or W < K or H < K:
        raise ValueError("raster smaller than the margin-cropped patch")
    xs = _axis_origins(W, patch_size, margin, step)
    ys = _axis_origins(H, patch_size, margin, step)  # distance of the window's bottom edge from the raster's bottom edge

    def owned(origins, extent):
        lo = [o + margin for o in origins]
        hi = [min(o + patch_size - margin, extent) for o in origins]
        return [(lo[i], min(hi[i], lo[i + 1]) if i + 1 < len(origins) else hi[i]) for i in range(len(origins))]

    ox, oy = owned(xs, W), owned(ys, H)
    rows = []
    for i, x in enumerate(xs):
        for j, yb in enumerate(ys):
            rows.append((x, H - (yb + patch_size), ox[i][0], ox[i][1], H - oy[j][1], H - oy[j][0]))
    return np.asarray(rows, dtype=np.int32)


class ZoneDetector:
    """``model`` over a whole raster: windows are cut, normalised, inferred, converted and stitched on the device; the
    raster goes up once as stored bytes and the result comes down once."""

    def __init__(self, model, config: dict):
        import ctypes as C
        self.model = model
        self.S = int(config["img_pixels_detection"])
        self.margin = int(config["margin"])
        self.output_type = config["output_type"]
        if self.output_type not in OUTPUT_TYPES:
            raise ValueError("The output type has not been interpreted.")
        if config.get("overlap_strat"):
            raise NotImplementedError("overlap strategies ('average', 'average_weights', 'max') are not built yet")
        self.stride = get_stride(config)[0]
        self.batch_size = int(config.get("batch_size", 4))
        self.channels = [int(c) for c in config["channels"]]
        norma = config["norma_task"][0]
        self.norm_type = norma["norm_type"] if norma["norm_type"] in ("custom", "scaling") else "scaling"
        means, stds = norma.get("norm_means", []), norma.get("norm_stds", [])
        if self.norm_type == "custom" and len(means) != len(stds):
            self.norm_type = "scaling"  # dataset.py:77-81
        n = len(self.channels)
        self._ch = (C.c_int * n)(*self.channels)
        self._means = (C.c_double * n)(*[float(m) for m in means[:n]]) if self.norm_type == "custom" else None
        self._stds = (C.c_double * n)(*[float(s) for s in stds[:n]]) if self.norm_type == "custom" else None
        self.n_classes = int(config["n_classes"])

    @torch.no_grad()
    def run(self, raster_u8: torch.Tensor) -> torch.Tensor:
        if not raster_u8.is_cuda or raster_u8.dtype != torch.uint8 or raster_u8.dim() != 3:
            raise RuntimeError("raster must be a (bands, H, W) uint8 tensor on the HIP device")
        raster_u8 = raster_u8.contiguous()
        bands, Hr, Wr = raster_u8.shape
        dev = raster_u8.device
        grid = torch.from_numpy(tile_grid((Wr, Hr), self.S, self.margin, self.stride)).to(dev)
        mode = OUTPUT_TYPES[self.output_type]
        out = (torch.zeros(2, Hr, Wr, dtype=torch.float32, device=dev) if mode == 0
               else torch.zeros(self.n_classes, Hr, Wr, dtype=torch.uint8, device=dev))
        from .data_feed import NORM_CODES
        for b0 in range(0, grid.shape[0], self.batch_size):
            tiles = grid[b0:b0 + self.batch_size].contiguous()
            B = tiles.shape[0]
            imgs = torch.empty(B, len(self.channels), self.S, self.S, dtype=torch.float32, device=dev)
            L.check(L.lib().flair_gather_tiles(L.ptr(raster_u8), bands, Hr, Wr, L.ptr(tiles), B, self.S, self._ch,
                                               len(self.channels), NORM_CODES[self.norm_type], self._means, self._stds,
                                               L.ptr(imgs), L.stream()), "flair_gather_tiles")
            if mode == 0 and hasattr(self.model, "predict_classes") and not self.model.training:
                # 'argmax' output: class and probability leave the head convolution's epilogue, the logits are never written
                if getattr(self.model, "classes", self.n_classes) != self.n_classes:
                    raise RuntimeError(f"model has {self.model.classes} classes, config says {self.n_classes}")
                preds, prob = self.model.predict_classes(imgs, want_prob=True)
                L.check(L.lib().flair_detect_stitch_preds(L.ptr(preds), L.ptr(prob), B, self.S, self.margin, L.ptr(tiles), L.ptr(out),
                                                          Hr, Wr, L.stream()), "flair_detect_stitch_preds")
                continue
            if hasattr(self.model, "forward_full"):
                # HuggingFace provider (SegFormer): `.logits` come at 1/4 of the tile size, which neither the margin crop nor
                # convert (compare.py:69-82) rescale; the x4 bilinear upsample (align_corners=False) the library itself applies
                # in front of its loss brings them to tile resolution first
                logits = self.model.forward_full(imgs)
            else:
                logits = self.model(imgs).float().contiguous()
            if logits.shape[1] != self.n_classes:
                raise RuntimeError(f"model returned {logits.shape[1]} classes, config says {self.n_classes}")
            L.check(L.lib().flair_detect_stitch(L.ptr(logits), B, self.n_classes, self.S, self.margin, mode, L.ptr(tiles),
                                                L.ptr(out), Hr, Wr, L.stream()), "flair_detect_stitch")
        return out


# ------------------------------------------------------------------------------------------------------------------
# Tile bookkeeping of the overlap strategies (src/zone_detect/test/tiles.py, test/pixel_operation.py): host integer /
# float64 arithmetic on small maps, mirrored for callers that schedule overlapping windows themselves.  Pinned by
# tests/golden/tiling_reference.json (the reference functions, imported directly).

def out_of_bounds(bigbox, box):
    """tiles.py:17-27 — per coordinate of ``box``: is it outside ANY of the four bounds of ``bigbox``
    (left, right, bottom, top), compared as the reference does (every coordinate against all four)."""
    left, right, bottom, top = bigbox
    return [bool(c < left or c > right or c < bottom or c > top) for c in box]


def get_tile_coord(start: int, end: int, limit: int, patch_size: int, stride: int):
    """tiles.py:30-51 — tile origins along one axis that intersect [start, end): multiples of ``stride`` below ``end``,
    the ones that would pass ``limit`` pulled back to limit - patch_size (ascending; the reference's order is a set's)."""
    last = limit - patch_size
    if last < 0:
        return []
    origins = {min(i, last) if i + patch_size > limit else i for i in range(0, end, stride)}
    return sorted(o for o in origins if o + patch_size > start and o < end)


def _axis_cover(lo: int, hi: int, limit: int, patch_size: int, stride: int):
    """how many tiles of one axis cover each position of [lo, hi)"""
    import numpy as np
    n = np.zeros(hi - lo, dtype=np.int64)
    for o in get_tile_coord(lo, hi, limit, patch_size, stride):
        a, b = max(o, lo), min(o + patch_size, hi)
        if b > a:
            n[a - lo:b - lo] += 1
    return n


def patch_overlap(image_size, patch_size: int, query_bounds, stride: int):
    """tiles.py:54-94 — number of tiles covering every pixel of the query rectangle (x_min, x_max, y_min, y_max); the
    tiles form a grid, so the count is the outer product of the per-axis coverages."""
    import numpy as np
    x_min, x_max, y_min, y_max = query_bounds
    cx = _axis_cover(x_min, x_max, image_size[0], patch_size, stride)
    cy = _axis_cover(y_min, y_max, image_size[1], patch_size, stride)
    return np.outer(cy, cx).astype(np.uint8)


def patch_weights(patch_size: int, sigma: float, mode: str):
    """tiles.py:97-108 — weight of a pixel by its Chebyshev distance d to the patch centre:
    'gaussian': exp(-d / d_max^2) / (2 sigma^2); anything else: exp(-d / d_max * sigma)."""
    import numpy as np
    c = patch_size // 2
    ax = np.abs(np.arange(patch_size) - c)
    dist = np.maximum(ax[:, None], ax[None, :])
    if mode == "gaussian":
        return np.exp(-dist / dist.max() ** 2) / (2 * sigma ** 2)
    return np.exp(-dist / dist.max() * sigma)


def total_weights(image_size, patch_size: int, query_bounds, stride: int, track_steps: bool = False):
    """tiles.py:111-168 — sum over the tiles intersecting the query of their 'exp' patch weights (sigma 0.5), float32."""
    import numpy as np
    x_min, x_max, y_min, y_max = query_bounds
    acc = np.zeros((y_max - y_min, x_max - x_min), dtype=np.float32)
    w = patch_weights(patch_size, sigma=0.5, mode="exp")
    steps = []
    for ty in get_tile_coord(y_min, y_max, image_size[1], patch_size, stride):
        ya, yb = max(ty, y_min), min(ty + patch_size, y_max)
        for tx in get_tile_coord(x_min, x_max, image_size[0], patch_size, stride):
            xa, xb = max(tx, x_min), min(tx + patch_size, x_max)
            if yb > ya and xb > xa:
                acc[ya - y_min:yb - y_min, xa - x_min:xb - x_min] += w[ya - ty:yb - ty, xa - tx:xb - tx]
                if track_steps:
                    steps.append(acc.copy())
    return acc, steps


def slice_pixels(img_size, patch_size: int, margin: int, stride: int):
    """pixel_operation.py:1-60 — margin-cropped boxes (x_min, x_max, y_min, y_max) of a regular grid with step
    ``stride``, plus a last row / column / corner flush with the image edge when the grid does not end there."""
    x_size, y_size = img_size
    k = patch_size - 2 * margin

    def axis(size):
        starts = [o for o in range(0, size + 1, stride) if o + k <= size]
        if size - k > 0 and (size - k) % stride != 0:
            starts.append(size - k)
        return starts

    xs, ys = axis(x_size), axis(y_size)
    return sorted({(x, x + k, y, y + k) for y in ys for x in xs})
