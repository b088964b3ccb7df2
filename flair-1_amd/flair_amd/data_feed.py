"""Device-side input contract of the step.

The reference prepares every tile on CPU DataLoader workers: rasterio read -> ``norm`` in float64 ->
float32, label raster -> ``C`` equality planes -> float32 one-hot (13x the label bytes over PCIe), and
albumentations flips/rot90 on HWC numpy (src/flair/data_loader.py:9-30,65-95; src/flair/tasks_utils.py:37-41).
Here a batch of STORED uint8 rasters goes to the GPU as is (5 + 1 bytes per pixel) and one launch of
``flair_feed_tiles`` emits what ``step()`` consumes: ``batch["img"]`` fp32 NCHW and ``batch["msk"]`` as the
uint8 class index the reference would get from ``argmax(one_hot, 1)`` (task_module.py:71).

No CPU fallback: tensors must live on a HIP device.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L

NORM_CODES = {"without": 0, "scaling": 1, "custom": 2}


def draw_d4(batch: int, generator: torch.Generator | None = None) -> torch.Tensor:
    """Per-sample draws of Compose([VerticalFlip(p=.5), HorizontalFlip(p=.5), RandomRotate90(p=.5)])
    (tasks_utils.py:37-41) packed as bit0 | bit1 << 1 | factor << 2; RandomRotate90 draws its factor
    uniformly from {0, 1, 2, 3} when it fires.  The draw STREAM is torch's, not albumentations' (absent
    here): same distribution, different sequence."""
    u = torch.rand(batch, 3, generator=generator)
    k = torch.randint(0, 4, (batch,), generator=generator)
    flags = (u[:, 0] < 0.5).to(torch.uint8) | ((u[:, 1] < 0.5).to(torch.uint8) << 1)
    flags |= torch.where(u[:, 2] < 0.5, k, torch.zeros_like(k)).to(torch.uint8) << 2
    return flags


def pack_d4(vflip, hflip, k) -> int:
    return int(bool(vflip)) | (int(bool(hflip)) << 1) | ((int(k) & 3) << 2)


class TileFeed:
    """Configured like the reference's datasets (``channels``, ``num_classes``, ``norm_type``, ``means``, ``stds``:
    data_loader.py:36-56,102-121) and called per batch instead of per tile."""

    def __init__(self, channels=(1, 2, 3, 4, 5), num_classes: int = 13, norm_type: str = "scaling", means=(), stds=(),
                 use_augmentations: bool = False, generator: torch.Generator | None = None):
        if norm_type not in NORM_CODES:
            print("Normalization argument should be 'scaling', 'custom' or 'without'.")
            raise SystemExit()
        if norm_type == "custom" and (len(means) != len(stds) or len(means) < len(channels)):
            print("If custom, provided normalization means and stds should be of same lenght.")
            raise SystemExit()
        self.channels = [int(c) for c in channels]
        self.num_classes = int(num_classes)
        self.norm_type = norm_type
        n = len(self.channels)
        self._ch = (C.c_int * n)(*self.channels)
        self._means = (C.c_double * n)(*[float(m) for m in list(means)[:n]]) if norm_type == "custom" else None
        self._stds = (C.c_double * n)(*[float(s) for s in list(stds)[:n]]) if norm_type == "custom" else None
        self.use_augmentations = bool(use_augmentations)
        self.generator = generator

    @classmethod
    def from_config(cls, config: dict, train: bool = True, generator=None) -> "TileFeed":
        """The arguments tasks_utils.get_data_module hands to flair_datamodule (tasks_utils.py:45-59)."""
        return cls(channels=config["channels"], num_classes=len(config["classes"]), norm_type=config["norm_type"],
                   means=config.get("norm_means", ()), stds=config.get("norm_stds", ()),
                   use_augmentations=bool(config.get("use_augmentation", False)) and train, generator=generator)

    def __call__(self, img_u8: torch.Tensor, msk_raw: torch.Tensor | None = None, d4: torch.Tensor | None = None,
                 mtd: torch.Tensor | None = None, ids=None) -> dict:
        """img_u8 (B, bands, H, W) uint8 and msk_raw (B, H, W) uint8 on the GPU -> the batch dict of
        data_loader.py:88-95 / :137-144 (``msk`` as class indices)."""
        if not img_u8.is_cuda:
            raise RuntimeError("TileFeed needs device tensors (no CPU fallback)")
        if img_u8.dtype != torch.uint8 or img_u8.dim() != 4:
            raise ValueError("img_u8 must be (B, bands, H, W) uint8")
        img_u8 = img_u8.contiguous()
        B, bands, H, W = img_u8.shape
        if max(self.channels) > bands or min(self.channels) < 1:
            raise ValueError(f"channels {self.channels} outside the raster's 1..{bands} bands")
        if d4 is None and self.use_augmentations:
            d4 = draw_d4(B, self.generator)
        if d4 is not None:
            d4 = d4.to(device=img_u8.device, dtype=torch.uint8).contiguous()
            if d4.numel() != B:
                raise ValueError("d4 needs one draw per sample")
        out = torch.empty(B, len(self.channels), H, W, dtype=torch.float32, device=img_u8.device)
        labels = None
        if msk_raw is not None:
            if msk_raw.dtype != torch.uint8 or tuple(msk_raw.shape) != (B, H, W) or msk_raw.device != img_u8.device:
                raise ValueError("msk_raw must be (B, H, W) uint8 on the image's device")
            msk_raw = msk_raw.contiguous()
            labels = torch.empty(B, H, W, dtype=torch.uint8, device=img_u8.device)
        L.check(L.lib().flair_feed_tiles(L.ptr(img_u8), L.ptr(msk_raw) if msk_raw is not None else None,
                                         L.ptr(d4) if d4 is not None else None, B, bands, H, W, self._ch, len(self.channels),
                                         NORM_CODES[self.norm_type], self._means, self._stds, self.num_classes, L.ptr(out),
                                         L.ptr(labels) if labels is not None else None, L.stream()), "flair_feed_tiles")
        batch = {"img": out}
        if mtd is not None:
            batch["mtd"] = mtd.to(device=out.device, dtype=torch.float32)
        if labels is not None:
            batch["msk"] = labels
        if ids is not None:
            batch["id"] = ids
        return batch
