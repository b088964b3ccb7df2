"""Oracle (TEST INFRASTRUCTURE): the per-pixel CE / argmax / IoU reduction of the reference.

Restates, in numpy (float64 / int64) and plain torch-CPU:

* ``segmentation_task_training.step``      /root/reference/src/flair/task_module.py:65-80
* ``segmentation_task_predict.predict_step``  task_module.py:206-213
* criterion ``nn.CrossEntropyLoss(weight)``  src/flair/tasks_utils.py:88-93
* torchmetrics-1.2.0 ``MulticlassJaccardIndex`` update/compute (task_module.py:36-51;
  third-party, absent here; published algorithm restated)
* ``class_IoU`` / ``overall_accuracy`` / ``class_precision`` / ``class_recall`` /
  ``class_fscore``  src/flair/metrics.py:10-40 and the weight-0 class removal
  metrics.py:77-88
* ``SGD(lr)`` without momentum  tasks_utils.py:95
* zone_detect ``inference`` + ``convert('argmax')``  src/zone_detect/compare.py:35,
  src/zone_detect/dataset.py:23-30

Pinned by tests/golden/metrics_*.json and step_*.npz, which were produced by running the
reference's own files in this container (tests/golden/make_golden.py).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- step()
def onehot_to_label(msk_onehot: torch.Tensor) -> torch.Tensor:
    """task_module.py:71 — argmax over the one-hot channel axis; an all-zero pixel -> 0."""
    return torch.argmax(msk_onehot, dim=1)


def cross_entropy_np(logits: np.ndarray, labels: np.ndarray, weight: np.ndarray | None = None):
    """float64 restatement of nn.CrossEntropyLoss(weight, reduction='mean').

    logits (B,C,H,W), labels (B,H,W) int.  Returns (loss, dlogits) with
    loss = sum_i w[y_i] * nll_i / sum_i w[y_i].
    """
    x = logits.astype(np.float64)
    B, C = x.shape[:2]
    m = x.max(axis=1, keepdims=True)
    e = np.exp(x - m)
    s = e.sum(axis=1, keepdims=True)
    logp = x - m - np.log(s)
    w = np.ones(C) if weight is None else np.asarray(weight, np.float64)
    wy = w[labels]
    onehot = np.zeros_like(x)
    np.put_along_axis(onehot, labels[:, None], 1.0, axis=1)
    nll = -(logp * onehot).sum(axis=1)
    den = wy.sum()
    loss = (wy * nll).sum() / den
    dlogits = (e / s - onehot) * (wy / den)[:, None]
    return loss, dlogits


def step_torch(logits: torch.Tensor, msk_onehot: torch.Tensor, weight: torch.Tensor | None = None):
    """task_module.py:70-80 on given logits: returns loss, preds (B,HW) int64, targets (B,HW) int32."""
    targets = torch.argmax(msk_onehot, dim=1)
    loss = F.cross_entropy(logits, targets, weight=weight)
    with torch.no_grad():
        proba = torch.softmax(logits, dim=1)
        preds = torch.argmax(proba, dim=1)
        preds = preds.flatten(start_dim=1)
        targets = targets.flatten(start_dim=1).type(torch.int32)
    return loss, preds, targets


def predict_torch(logits: torch.Tensor) -> torch.Tensor:
    """task_module.py:211-212: argmax(softmax(logits)) -> (B,H,W) int64."""
    return torch.argmax(torch.softmax(logits, dim=1), dim=1)


def detect_convert_argmax(probs: np.ndarray) -> np.ndarray:
    """zone_detect/dataset.py:23-30 on one (C,H,W) probability tile -> (2,H,W) float32."""
    a = np.argmax(probs, axis=0).astype(np.uint8)[None]
    m = np.max(probs, axis=0).astype(np.float32)[None]
    return np.concatenate([a, m], axis=0)


# ----------------------------------------------------------------------------- confusion matrix / Jaccard
def confusion_matrix_np(target: np.ndarray, pred: np.ndarray, num_classes: int) -> np.ndarray:
    """torchmetrics ``_multiclass_confusion_matrix_update``: bincount(target*C + pred) -> [target, pred].

    Also what sklearn.metrics.confusion_matrix(labels=range(C)) returns (metrics.py:67-71);
    pairs with an out-of-range member are dropped like sklearn does.
    """
    t = np.asarray(target).reshape(-1).astype(np.int64)
    p = np.asarray(pred).reshape(-1).astype(np.int64)
    ok = (t >= 0) & (t < num_classes) & (p >= 0) & (p < num_classes)
    return np.bincount(t[ok] * num_classes + p[ok], minlength=num_classes ** 2).reshape(num_classes, num_classes)


def jaccard_from_confmat(confmat: np.ndarray, average: str | None = "weighted"):
    """torchmetrics-1.2.0 ``_jaccard_index_reduce`` (multiclass): 'weighted' | 'macro' | None."""
    cm = confmat.astype(np.float64)
    num = np.diag(cm)
    denom = cm.sum(0) + cm.sum(1) - num
    jac = np.where(denom != 0, num / np.where(denom == 0, 1, denom), 0.0)
    if average is None or average == "none":
        return jac
    if average == "macro":
        # torchmetrics macro ignores classes absent from both target and pred
        present = denom != 0
        return jac[present].mean() if present.any() else 0.0
    if average == "weighted":
        support = cm.sum(1)
        tot = support.sum()
        return float((jac * support / tot).sum()) if tot > 0 else 0.0
    raise ValueError(average)


# ----------------------------------------------------------------------------- metrics.py family
def overall_accuracy(npcm):  # metrics.py:10-12
    return 100 * np.trace(npcm) / npcm.sum()


def class_IoU(npcm, n_class=None):  # metrics.py:15-22
    with np.errstate(divide="ignore", invalid="ignore"):
        ious = 100 * np.diag(npcm) / (np.sum(npcm, axis=1) + np.sum(npcm, axis=0) - np.diag(npcm))
    ious[np.isnan(ious)] = 0
    return ious, np.mean(ious)


def class_precision(npcm):  # metrics.py:25-28
    with np.errstate(divide="ignore", invalid="ignore"):
        precision = 100 * np.diag(npcm) / np.sum(npcm, axis=0)
    precision[np.isnan(precision)] = 0
    return precision, np.mean(precision)


def class_recall(npcm):  # metrics.py:31-34
    with np.errstate(divide="ignore", invalid="ignore"):
        recall = 100 * np.diag(npcm) / np.sum(npcm, axis=1)
    recall[np.isnan(recall)] = 0
    return recall, np.mean(recall)


def class_fscore(precision, recall):  # metrics.py:37-40
    with np.errstate(divide="ignore", invalid="ignore"):
        fscore = 2 * (precision * recall) / (precision + recall)
    fscore[np.isnan(fscore)] = 0
    return fscore, np.mean(fscore)


def offline_metrics(sum_confmat: np.ndarray, weights) -> dict:
    """metrics.py:75-108: drop weight-0 classes, then the five averages + per-class lists."""
    weights = np.asarray(weights)
    unused = np.where(weights == 0)[0]
    cm = np.delete(np.delete(sum_confmat, unused, axis=0), unused, axis=1)
    iou, miou = class_IoU(cm, len(np.nonzero(weights)[0]))
    oa = overall_accuracy(cm)
    p, mp = class_precision(cm)
    r, mr = class_recall(cm)
    f, mf = class_fscore(p, r)
    return {
        "Avg_metrics_name": ["mIoU", "Overall Accuracy", "Fscore", "Precision", "Recall"],
        "Avg_metrics": [miou, oa, mf, mp, mr],
        "per_class_iou": list(iou),
        "per_class_fscore": list(f),
        "per_class_precision": list(p),
        "per_class_recall": list(r),
    }


# ----------------------------------------------------------------------------- optimiser
def sgd_step_(params, lr: float):
    """tasks_utils.py:95 — torch.optim.SGD(lr): w <- w - lr * g (no momentum / decay)."""
    with torch.no_grad():
        for p in params:
            if p.grad is not None:
                p.add_(p.grad, alpha=-lr)
