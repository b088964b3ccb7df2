"""Mirror of the hot-path parts of /root/reference/src/flair/tasks_utils.py.

  * ``get_segmentation_module(config, stage)``  tasks_utils.py:64-121 — same asserts, criterion
    (``use_weights`` -> class weights from the YAML ``classes`` table), plain SGD, ReduceLROnPlateau(mode=min,
    factor .5, patience 10, cooldown 4, min_lr 1e-7), but building the HIP-backed modules of this package.
  * ``get_tile_feed(config)``                   the arguments of ``get_data_module`` (tasks_utils.py:16-60) applied
    to the device feed (``flair_amd.data_feed.TileFeed``) instead of CPU DataLoader workers.
  * ``parsing_metadata(image_path_list, config)`` tasks_utils.py:158-213 — the 45-d metadata encoding
    (32 positional + 1 altitude + 2 camera + 4 year + 6 cyclic date/time), host-side float64 like the reference.

File gathering (``gather_paths``) and the Lightning DataModule are outside the hot path.
"""
from __future__ import annotations

import json
import math

import torch
from torch import nn
from torch.optim.lr_scheduler import ReduceLROnPlateau

from .data_feed import TileFeed
from .head import FusedCrossEntropyLoss
from .model import FLAIR_ModelFactory
from .task_module import segmentation_task_predict, segmentation_task_training


def get_tile_feed(config, train: bool = True, generator=None) -> TileFeed:
    assert isinstance(config, dict), "config must be a dictionary"
    assert isinstance(config["use_augmentation"], bool), "use_augmentation must be a boolean"
    assert isinstance(config["use_metadata"], bool), "use_metadata must be a boolean"
    return TileFeed.from_config(config, train=train, generator=generator)


def get_segmentation_module(config, stage: str = "train", compute_dtype=None):
    assert stage in ["train", "predict"], "stage must be either 'train' or 'predict'"
    assert config["model_framework"]["model_provider"] in ["HuggingFace", "SegmentationModelsPytorch"], \
        "model_provider should be either HuggingFace or SegmentationModelsPytorch"
    model = FLAIR_ModelFactory(config) if compute_dtype is None else FLAIR_ModelFactory(config, compute_dtype=compute_dtype)
    if stage == "train":
        if config["use_weights"]:
            with torch.no_grad():
                class_weights = torch.FloatTensor([config["classes"][i][0] for i in config["classes"]])
            criterion = FusedCrossEntropyLoss(weight=class_weights)
        else:
            criterion = FusedCrossEntropyLoss()
        optimizer = torch.optim.SGD(model.parameters(), lr=config["learning_rate"])
        scheduler = ReduceLROnPlateau(optimizer=optimizer, mode="min", factor=0.5, patience=10, cooldown=4, min_lr=1e-7)
        return segmentation_task_training(model=model, class_infos=config["classes"], criterion=criterion,
                                          optimizer=optimizer, scheduler=scheduler, use_metadata=config["use_metadata"])
    return segmentation_task_predict(model=model, num_classes=len(config["classes"]), use_metadata=config["use_metadata"])


_YEARS = ("2018", "2019", "2020", "2021")
_MAX_ALTI = 3164.9099121094


def _positional(value: float, n_freq: int = 8, d: int = 16):
    """sin/cos pairs at 8 frequencies 1 / (10e7 ** (2 i / 16)) of value / 10e7."""
    v = value / 10e7
    out = []
    for i in range(n_freq):
        f = 1 / (10e7 ** (2 * float(i) / d))
        out += [math.sin(v * f), math.cos(v * f)]
    return out


def encode_metadata_record(rec: dict) -> list:
    enc = _positional(rec["patch_centroid_x"]) + _positional(rec["patch_centroid_y"])
    enc.append((rec["patch_centroid_z"] - 0) / (_MAX_ALTI - 0))
    enc += [1, 0] if "UCE" in rec["camera"] else [0, 1]
    year, month, day = rec["date"].split("-")
    if year not in _YEARS:
        raise NameError(f"year {year!r} has no encoding (the reference knows 2018-2021 only)")
    enc += [1 if year == y else 0 for y in _YEARS]
    hours, minutes = rec["time"].split("h")
    angles = (2 * math.pi * (int(month) - 1 / 12),  # the reference's month term, kept as written there
              2 * math.pi * (int(day) / 31),
              2 * math.pi * ((int(hours) * 3600 + int(minutes) * 60) / 86400))
    for a in angles:
        enc += [(math.sin(a) + 1) / 2, (math.cos(a) + 1) / 2]
    return enc


def parsing_metadata(image_path_list, config):
    with open(config["paths"]["path_metadata_aerial"], "r") as f:
        metadata_dict = json.load(f)
    return [encode_metadata_record(metadata_dict[img.split("/")[-1][:-4]]) for img in image_path_list]
