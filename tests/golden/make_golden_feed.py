#!/usr/bin/env python3
"""Golden vectors for the rows either side of the hot path, produced by
running the REFERENCE's own files.  Build container only (needs /root/reference, read-only):

    python tests/golden/make_golden_feed.py

Executed from the reference, unchanged, loaded by path:
  * src/flair/data_loader.py     norm ('custom', 'without'), fit_dataset.__getitem__ / read_msk,
                                 predict_dataset.__getitem__
  * src/flair/tasks_utils.py     parsing_metadata (45-d metadata encoding)
  * src/zone_detect/dataset.py   convert ('argmax', 'class_prob')
  * src/zone_detect/test/tiles.py, test/pixel_operation.py   (import without any stand-in) get_stride, out_of_bounds,
                                 get_tile_coord, patch_overlap, patch_weights, total_weights, slice_pixels

Absent third-party packages are replaced in ``sys.modules`` by stand-ins that carry NO arithmetic of the
path: rasterio (``open`` hands back arrays this script made; nothing is decoded), geopandas, albumentations,
pytorch_lightning, torchmetrics, transformers (names only).  skimage's ``img_as_float`` RAISES in the
stand-in: the 'scaling' normalisation is therefore NOT covered by a golden (parity unpinned, restated from
skimage's published dtype conversion in oracle/data_feed.py); neither is the albumentations draw sequence.

Outputs are DATA only (inputs + expected outputs); no reference source text is stored.
"""
import contextlib
import importlib.util
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

RASTERS = {}  # fake "files": path -> (bands, H, W) uint8 array


class _FakeRaster:
    def __init__(self, arr):
        self.arr = arr

    def read(self, indexes=None):
        if indexes is None:
            return self.arr.copy()
        return np.stack([self.arr[i - 1] for i in indexes], axis=0)


@contextlib.contextmanager
def _fake_open(path, *a, **k):
    yield _FakeRaster(RASTERS[path])


def _unavailable(*a, **k):
    raise RuntimeError("third-party arithmetic absent in this container: no golden for this branch")


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mod("rasterio", open=_fake_open, windows=mod("rasterio.windows"), enums=mod("rasterio.enums", Resampling=object))
    mod("geopandas", GeoDataFrame=object)
    mod("skimage", img_as_float=_unavailable, util=mod("skimage.util", img_as_float=_unavailable))
    names = {n: (lambda *a, **k: None) for n in ("Compose", "VerticalFlip", "HorizontalFlip", "RandomRotate90")}
    mod("albumentations", **names)
    mod("pytorch_lightning", LightningModule=nn.Module, LightningDataModule=object,
        utilities=mod("pytorch_lightning.utilities"))
    mod("pytorch_lightning.utilities.rank_zero", rank_zero_only=lambda f: f)
    mod("torchmetrics")
    mod("torchmetrics.classification", MulticlassJaccardIndex=type("MulticlassJaccardIndex", (), {}))
    mod("torchmetrics.aggregation", MeanMetric=type("MeanMetric", (), {}))
    mod("transformers", AutoModelForSemanticSegmentation=object, AutoConfig=object)
    mod("segmentation_models_pytorch", create_model=_unavailable)


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


MEANS = [105.08, 110.87, 101.82, 106.38, 53.26]  # configs/flair-1-config.yaml:44
STDS = [52.17, 45.38, 44, 39.69, 79.3]           # configs/flair-1-config.yaml:45


def golden_feed(dl):
    rng = np.random.default_rng(2022)
    H = 48
    out = {}
    # every uint8 value in every band at least once (the normalisation is a per-band function of the byte)
    img = rng.integers(0, 256, size=(5, H, H), dtype=np.uint8)
    img[:, 0, :] = np.arange(H, dtype=np.uint8)[None]
    img.reshape(5, -1)[:, :256] = np.arange(256, dtype=np.uint8)[None]
    raw = rng.integers(0, 22, size=(1, H, H), dtype=np.uint8)  # 0 (-> 255 after -1) and > C present
    RASTERS["t/IMG_000001.tif"] = img
    RASTERS["t/MSK_000001.tif"] = raw
    files = {"IMG": ["t/IMG_000001.tif"], "MSK": ["t/MSK_000001.tif"]}
    for C in (13, 19):
        ds = dl.fit_dataset(files, channels=[1, 2, 3, 4, 5], num_classes=C, use_metadata=False, use_augmentations=None,
                            norm_type="custom", means=MEANS, stds=STDS)
        s = ds[0]
        assert s["img"].dtype == torch.float32 and s["msk"].dtype == torch.float32
        out[f"custom_img_c{C}"] = s["img"].numpy()
        out[f"msk_onehot_c{C}"] = s["msk"].numpy().astype(np.uint8)
    ds = dl.fit_dataset(files, channels=[3, 1, 5], num_classes=13, use_metadata=False, use_augmentations=None,
                        norm_type="custom", means=[1.5, 2.25, 100.0], stds=[3.0, 0.7, 51.3])
    out["custom_img_ch315"] = ds[0]["img"].numpy()
    ds = dl.predict_dataset(files, channels=[1, 2, 3, 4, 5], num_classes=13, use_metadata=False, norm_type="without")
    s = ds[0]
    assert sorted(s.keys()) == ["id", "img"]
    out["without_img"] = s["img"].numpy()
    np.savez_compressed(os.path.join(HERE, "feed_reference.npz"), img_u8=img, msk_raw=raw[0], means=np.array(MEANS),
                        stds=np.array(STDS, dtype=np.float64), **out)
    print("feed:", {k: v.shape for k, v in out.items()})


def golden_metadata_encoding(tu):
    records = {
        "IMG_061946": {"patch_centroid_x": 489353.59, "patch_centroid_y": 6359290.71, "patch_centroid_z": 121.51,
                       "camera": "UCE-M3-f120-s06", "date": "2020-08-30", "time": "13h30"},
        "IMG_000007": {"patch_centroid_x": 1020000.0, "patch_centroid_y": 6100000.5, "patch_centroid_z": 3164.9099121094,
                       "camera": "UCXp-f100", "date": "2018-01-01", "time": "9h05"},
        "IMG_123456": {"patch_centroid_x": 99999.99, "patch_centroid_y": 7100123.25, "patch_centroid_z": 0,
                       "camera": "DMCIIe_UCE", "date": "2021-12-31", "time": "23h59"},
        "IMG_000019": {"patch_centroid_x": 651000.0, "patch_centroid_y": 6860000.0, "patch_centroid_z": 35.0,
                       "camera": "xyz", "date": "2019-06-15", "time": "0h00"},
    }
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "flair-1_metadata_aerial.json")
        json.dump(records, open(p, "w"))
        paths = [f"/data/D0{i}_2020/Z1_UU/img/{k}.tif" for i, k in enumerate(records)]
        enc = tu.parsing_metadata(paths, {"paths": {"path_metadata_aerial": p}})
    enc = np.asarray(enc, dtype=np.float64)
    assert enc.shape == (4, 45), enc.shape
    with open(os.path.join(HERE, "metadata_encoding_reference.json"), "w") as f:
        json.dump({"records": records, "paths": paths, "encoding": enc.tolist()}, f)
    print("metadata encoding:", enc.shape, float(enc.sum()))


def golden_convert(zd):
    g = torch.Generator().manual_seed(19)
    logits = torch.randn(19, 40, 40, generator=g) * 3.0
    probs = torch.softmax(logits, dim=0).numpy()  # compare.py:35 (softmax over the class axis), one tile
    am = zd.convert(probs.copy(), "argmax")
    cp = zd.convert(probs.copy(), "class_prob")
    assert am.dtype == np.float32 and am.shape == (2, 40, 40) and cp.dtype == np.uint8
    # exact ties: argmax keeps the FIRST maximum
    tie = np.zeros((4, 2, 2), dtype=np.float32)
    tie[1] = tie[3] = 0.5
    am_tie = zd.convert(tie.copy(), "argmax")
    np.savez_compressed(os.path.join(HERE, "detect_convert_reference.npz"), logits=logits.numpy(), probs=probs,
                        argmax=am, class_prob=cp, tie=tie, argmax_tie=am_tie)
    print("convert:", am.shape, cp.shape, am_tie[0].ravel())


def golden_tiling(tiles, pixop):
    cases = [
        {"img_size": [5, 5], "patch_size": 3, "margin": 0, "stride": 2, "query": [0, 5, 0, 5]},      # the reference's toy
        {"img_size": [10, 10], "patch_size": 4, "margin": 0, "stride": 2, "query": [0, 10, 0, 10]},
        {"img_size": [10, 10], "patch_size": 4, "margin": 1, "stride": 3, "query": [2, 9, 1, 7]},
        {"img_size": [64, 48], "patch_size": 16, "margin": 2, "stride": 8, "query": [5, 40, 7, 33]},
        {"img_size": [33, 70], "patch_size": 16, "margin": 4, "stride": 12, "query": [0, 33, 16, 70]},
    ]
    out = []
    for c in cases:
        size, P, m, st, q = tuple(c["img_size"]), c["patch_size"], c["margin"], c["stride"], c["query"]
        tw, _ = tiles.total_weights(size, P, q, st)
        out.append({**c,
                    "patch_overlap": tiles.patch_overlap(size, P, q, st).tolist(),
                    "total_weights": np.asarray(tw, dtype=np.float64).tolist(),
                    "tile_coord_x": sorted(tiles.get_tile_coord(q[0], q[1], size[0], P, st)),
                    "slice_pixels": [list(t) for t in pixop.slice_pixels(size, P, m, st)]})
    extra = {
        "patch_weights_exp_7": tiles.patch_weights(7, 0.5, "exp").tolist(),
        "patch_weights_gauss_8": tiles.patch_weights(8, 0.5, "gaussian").tolist(),
        "get_stride_default": tiles.get_stride({"img_pixels_detection": 512, "margin": 128}),
        "get_stride_overlap": tiles.get_stride({"img_pixels_detection": 512, "margin": 128, "overlap_strat": True,
                                                "strategies": {"tiling": {"stride_range": [0.25, 0.5, 0.75]}}}),
        "out_of_bounds": [[list(b), list(x), tiles.out_of_bounds(list(b), list(x))]
                          for b, x in (((0, 10, 0, 10), (0, 5, 0, 5)), ((0, 10, 0, 10), (-1, 5, 2, 11)),
                                       ((100.0, 200.0, 50.0, 80.0), (120.0, 130.0, 60.0, 70.0)))],
    }
    with open(os.path.join(HERE, "tiling_reference.json"), "w") as f:
        json.dump({"cases": out, **extra}, f)
    print("tiling:", len(out), "cases")


if __name__ == "__main__":
    _install_stubs()
    sys.path.insert(0, REF)  # the reference imports itself as `src.flair...`
    dl = _load("ref_flair_data_loader", "src/flair/data_loader.py")
    golden_feed(dl)
    tu = _load("ref_flair_tasks_utils", "src/flair/tasks_utils.py")
    golden_metadata_encoding(tu)
    zd = _load("ref_zone_detect_dataset", "src/zone_detect/dataset.py")
    golden_convert(zd)
    golden_tiling(_load("ref_zone_detect_tiles", "src/zone_detect/test/tiles.py"),
                  _load("ref_zone_detect_pixop", "src/zone_detect/test/pixel_operation.py"))
