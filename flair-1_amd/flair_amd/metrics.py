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


def metrics_from_confmat(sum_confmat: np.ndarray, classes: dict) -> dict:
    """metrics.py:76-108: drop the weight-0 classes, then the dict that becomes metrics.json."""
    weights = np.array([classes[i][0] for i in classes])
    unused_classes = np.where(weights == 0)[0]
    confmat_cleaned = np.delete(np.delete(sum_confmat, unused_classes, axis=0), unused_classes, axis=1)
    per_c_ious, avg_ious = class_IoU(confmat_cleaned, len(np.nonzero(weights)[0]))
    ovr_acc = overall_accuracy(confmat_cleaned)
    per_c_precision, avg_precison = class_precision(confmat_cleaned)
    per_c_recall, avg_recall = class_recall(confmat_cleaned)
    per_c_fscore, avg_fscore = class_fscore(per_c_precision, per_c_recall)
    return {
        "Avg_metrics_name": ["mIoU", "Overall Accuracy", "Fscore", "Precision", "Recall"],
        "Avg_metrics": [avg_ious, ovr_acc, avg_fscore, avg_precison, avg_recall],
        "classes": list(np.array([classes[i][1] for i in classes])[np.nonzero(weights)[0]]),
        "per_class_iou": list(per_c_ious),
        "per_class_fscore": list(per_c_fscore),
        "per_class_precision": list(per_c_precision),
        "per_class_recall": list(per_c_recall),
    }


def metrics(config: dict, path_preds, remove_preds: bool = False, device="cuda", chunk_tiles: int = 64):
    """Same inputs and outputs as the reference's ``metrics`` (metrics.py:44-125): test CSV -> truth rasters,
    ``PRED_<name>`` rasters under ``path_preds`` -> ``<out>/metrics/{confmat.npy, metrics.json}``; returns the dict."""
    import pandas as pd
    from PIL import Image

    path_preds = Path(path_preds)
    gt_csv = pd.read_csv(config["paths"]["test_csv"], header=None)
    truth_images = gt_csv.iloc[:, 0].to_list()
    truth_msks = gt_csv.iloc[:, 1].to_list()
    preds_msks = [Path(path_preds.as_posix(), "PRED_" + i.split("/")[-1]).as_posix() for i in truth_images]
    assert len(truth_msks) == len(preds_msks), "[WARNING !] mismatch number of predictions and test files."
    print("-- Calculating metrics --")
    acc = MaskConfusion(len(config["classes"]), device=device)
    pend_t, pend_p = [], []

    def flush():
        if pend_t:
            t = torch.from_numpy(np.concatenate(pend_t)).to(device, non_blocking=True)
            p = torch.from_numpy(np.concatenate(pend_p)).to(device, non_blocking=True)
            acc.update(t, p)
            pend_t.clear()
            pend_p.clear()

    for u in range(len(truth_msks)):
        try:
            target = _as_u8(np.asarray(Image.open(truth_msks[u])), truth=True)
            preds = _as_u8(np.asarray(Image.open(preds_msks[u])), truth=False)
            if target.size != preds.size:
                raise ValueError("Found input variables with inconsistent numbers of samples")
            pend_t.append(target.reshape(-1))
            pend_p.append(preds.reshape(-1))
        except Exception as e:  # noqa: BLE001 (the reference reports and skips the pair)
            print(f"Error at index {u}: {e}")
        if len(pend_t) >= chunk_tiles:
            flush()
    flush()
    sum_confmat = acc.compute()
    out = metrics_from_confmat(sum_confmat, config["classes"])

    out_folder_metrics = Path("/".join(path_preds.as_posix().split("/")[:-1]), "metrics")
    out_folder_metrics.mkdir(exist_ok=True, parents=True)
    np.save(out_folder_metrics.as_posix() + "/confmat.npy", sum_confmat)
    json.dump(out, open(out_folder_metrics / Path("metrics.json"), "w"))

    print("")
    print("Global Metrics: ")
    print("-" * 90)
    for metric_name, metric_value in zip(out["Avg_metrics_name"], out["Avg_metrics"]):
        print(f"{metric_name:<20s} {metric_value:<20.4f}")
    print("-" * 90 + "\n\n")
    if remove_preds:
        import shutil
        shutil.rmtree(path_preds)
    return out
