"""Mirror of /root/reference/src/flair/metrics.py with the per-tile confusion matrices on the GPU (SURVEY.md §8a-10,
§8f f2).

The reference decodes each truth/prediction TIFF pair and calls ``sklearn.confusion_matrix`` per tile on one CPU
process (metrics.py:60-75).  Here decoded uint8 rasters are shipped in chunks and ``flair_confmat_masks`` folds the
``- 1`` label shift, the ``labels=range(C)`` filter and the sum over tiles into one int64 C x C matrix on the device;
what follows (weight-0 class removal, the IoU / OA / precision / recall / F-score family, ``metrics.json`` and
``confmat.npy``) is float64 arithmetic on that C x C matrix and stays on the host, in the reference's operation order.
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import torch

from . import _lib as L


def _ratio_percent(num, den):
    """100 * num / den with 0/0 -> 0 (the reference zeroes NaNs only)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        out = 100 * num / den
    out[np.isnan(out)] = 0
    return out


def overall_accuracy(npcm):  # metrics.py:10-12
    return 100 * np.trace(npcm) / np.sum(npcm)


def class_IoU(npcm, n_class=None):  # metrics.py:15-22
    tp = np.diag(npcm)
    ious = _ratio_percent(tp, np.sum(npcm, axis=1) + np.sum(npcm, axis=0) - tp)
    return ious, np.mean(ious)


def class_precision(npcm):  # metrics.py:25-28
    precision = _ratio_percent(np.diag(npcm), np.sum(npcm, axis=0))
    return precision, np.mean(precision)


def class_recall(npcm):  # metrics.py:31-34
    recall = _ratio_percent(np.diag(npcm), np.sum(npcm, axis=1))
    return recall, np.mean(recall)


def class_fscore(precision, recall):  # metrics.py:37-40
    with np.errstate(divide="ignore", invalid="ignore"):
        fscore = 2 * (precision * recall) / (precision + recall)
    fscore[np.isnan(fscore)] = 0
    return fscore, np.mean(fscore)


def _as_u8(a: np.ndarray, truth: bool) -> np.ndarray:
    """Any integer raster as bytes without changing which pairs count: values the class filter drops anyway
    (negative, > 255) go to a byte that is dropped too (for truth, a stored 0 becomes 255 after the -1 shift)."""
    if a.dtype == np.uint8:
        return a
    if a.dtype.kind not in "iu":
        raise TypeError("mask rasters must hold integers")
    bad = (a < 0) | (a > 255)
    return np.where(bad, 0 if truth else 255, a).astype(np.uint8)


class MaskConfusion:
    """Device accumulator of sum_tiles confusion_matrix(truth - 1, pred, labels=range(C)) (metrics.py:60-75)."""

    def __init__(self, num_classes: int, device="cuda", truth_offset: int = -1):
        self.C = int(num_classes)
        self.truth_offset = int(truth_offset)
        self.confmat = torch.zeros(self.C, self.C, dtype=torch.int64, device=device)

    def update(self, truth_raw: torch.Tensor, pred: torch.Tensor) -> None:
        if truth_raw.dtype != torch.uint8 or pred.dtype != torch.uint8:
            raise ValueError("rasters must be uint8")
        if truth_raw.numel() != pred.numel():
            raise ValueError("[WARNING !] mismatch number of predictions and test files.")
        truth_raw, pred = truth_raw.contiguous(), pred.contiguous()
        L.check(L.lib().flair_confmat_masks(L.ptr(truth_raw), L.ptr(pred), truth_raw.numel(), self.C, self.truth_offset,
                                            L.ptr(self.confmat), L.stream()), "flair_confmat_masks")

    def compute(self) -> np.ndarray:
        return self.confmat.cpu().numpy()


# metrics.json layout (metrics.py:90-108): the five averages in this order, then the kept class names and four per-class lists
_AVG_NAMES = ("mIoU", "Overall Accuracy", "Fscore", "Precision", "Recall")


def metrics_from_confmat(sum_confmat: np.ndarray, classes: dict) -> dict:
    """What the reference derives from the summed confusion matrix (metrics.py:76-108): classes of weight 0 leave the matrix
    (rows and columns), the IoU / OA / F-score / precision / recall family is computed on the rest."""
    keep = [n for n, k in enumerate(classes) if classes[k][0] != 0]
    cm = np.asarray(sum_confmat)[np.ix_(keep, keep)]
    iou, miou = class_IoU(cm, len(keep))
    prec, mprec = class_precision(cm)
    rec, mrec = class_recall(cm)
    f1, mf1 = class_fscore(prec, rec)
    names = [classes[k][1] for k in classes]
    out = {"Avg_metrics_name": list(_AVG_NAMES), "Avg_metrics": [miou, overall_accuracy(cm), mf1, mprec, mrec],
           "classes": list(np.array(names)[keep])}
    for key, per_class in (("iou", iou), ("fscore", f1), ("precision", prec), ("recall", rec)):
        out["per_class_" + key] = list(per_class)
    return out


def _raster_pairs(test_csv, path_preds: Path):
    """(index, truth path, prediction path) per row of the test CSV: column 0 names the image (its basename gives
    PRED_<name>), column 1 the truth mask (metrics.py:45-55)."""
    import pandas as pd
    rows = pd.read_csv(test_csv, header=None)
    for u, (img, msk) in enumerate(zip(rows.iloc[:, 0], rows.iloc[:, 1])):
        yield u, msk, (path_preds / ("PRED_" + str(img).split("/")[-1])).as_posix()


def metrics(config: dict, path_preds, remove_preds: bool = False, device="cuda", chunk_tiles: int = 64):
    """Same inputs and outputs as the reference's ``metrics`` (metrics.py:44-125): test CSV -> truth rasters,
    ``PRED_<name>`` rasters under ``path_preds`` -> ``<out>/metrics/{confmat.npy, metrics.json}``; returns the dict.
    A pair that cannot be read or whose sizes differ is reported and skipped, like the reference's try/except."""
    from PIL import Image

    path_preds = Path(path_preds)
    print("-- Calculating metrics --")
    acc = MaskConfusion(len(config["classes"]), device=device)
    pending = ([], [])

    def flush():
        if pending[0]:
            acc.update(*(torch.from_numpy(np.concatenate(side)).to(device, non_blocking=True) for side in pending))
            pending[0].clear()
            pending[1].clear()

    for u, truth_path, pred_path in _raster_pairs(config["paths"]["test_csv"], path_preds):
        try:
            target = _as_u8(np.asarray(Image.open(truth_path)), truth=True)
            preds = _as_u8(np.asarray(Image.open(pred_path)), truth=False)
            if target.size != preds.size:
                raise ValueError("Found input variables with inconsistent numbers of samples")
        except Exception as e:  # noqa: BLE001
            print(f"Error at index {u}: {e}")
            continue
        pending[0].append(target.reshape(-1))
        pending[1].append(preds.reshape(-1))
        if len(pending[0]) >= chunk_tiles:
            flush()
    flush()
    sum_confmat = acc.compute()
    out = metrics_from_confmat(sum_confmat, config["classes"])

    folder = path_preds.parent / "metrics"
    folder.mkdir(exist_ok=True, parents=True)
    np.save(folder / "confmat.npy", sum_confmat)
    with open(folder / "metrics.json", "w") as f:
        json.dump(out, f)
    rule = "-" * 90
    print("\nGlobal Metrics: \n" + rule)
    for name, value in zip(out["Avg_metrics_name"], out["Avg_metrics"]):
        print(f"{name:<20s} {value:<20.4f}")
    print(rule + "\n\n")
    if remove_preds:
        import shutil
        shutil.rmtree(path_preds)
    return out
