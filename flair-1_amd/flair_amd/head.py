"""Drop-in criterion / metric objects backed by the fused HIP head (csrc/ce_head.hip).

The reference injects the criterion and builds its metrics here:
  * criterion  ``nn.CrossEntropyLoss(weight=...)``            /root/reference/src/flair/tasks_utils.py:88-93
  * ``MulticlassJaccardIndex`` / ``MeanMetric`` (torchmetrics 1.2.0, absent in this image)
                                                                src/flair/task_module.py:36-51,84-85,106-108
``FusedCrossEntropyLoss`` has the call signature of ``nn.CrossEntropyLoss`` for (B,C,H,W) logits and
(B,H,W) integer targets, so ``tasks_utils.get_segmentation_module`` can hand it to the reference's own
LightningModule unchanged; the same kernel launch also yields predictions and the confusion matrix,
which ``flair_amd.task_module`` uses to skip the separate softmax/argmax/bincount passes.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, weight, owner):
        loss, dl, preds, _ = ops.ce_head(logits.detach().float().contiguous(), labels, weight, want_dlogits=True,
                                         want_preds="i64", confmat=owner._confmat_sink)
        owner.last_preds = preds
        ctx.save_for_backward(dl)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None, None


class FusedCrossEntropyLoss(nn.Module):
    """``nn.CrossEntropyLoss(weight, reduction='mean')`` for segmentation logits, one HIP pass.

    After each call ``last_preds`` holds argmax(softmax(logits)) (B,H,W) int64 — the reference's
    ``preds`` of task_module.py:75-76 — computed by the same kernel.
    """

    def __init__(self, weight=None):
        super().__init__()
        self.register_buffer("weight", None if weight is None else torch.as_tensor(weight, dtype=torch.float32))
        self.last_preds = None
        self._confmat_sink = None

    def forward(self, logits, target):
        if target.dtype == torch.float32 and target.dim() == 4:
            labels = target.contiguous()  # fp32 one-hot (the reference's batch["msk"] layout)
        else:
            labels = target.contiguous()
            if labels.dtype not in (torch.uint8, torch.int32, torch.int64):
                labels = labels.long()
        w = None if self.weight is None else self.weight.to(logits.device)
        return _CEFn.apply(logits, labels, w, self)


class MulticlassJaccardIndex(nn.Module):
    """torchmetrics-1.2.0 ``MulticlassJaccardIndex(num_classes, average)`` semantics on the device:
    update = bincount(target*C + pred) into an int64 confusion matrix; compute = Jaccard reduce."""

    def __init__(self, num_classes, average="macro", **_):
        super().__init__()
        self.num_classes, self.average = num_classes, average
        # metric STATE, not a parameter of the task: torchmetrics keeps its states out of state_dict() (persistent=False),
        # so reference checkpoints neither contain nor expect these keys
        self.register_buffer("confmat", torch.zeros(num_classes, num_classes, dtype=torch.int64), persistent=False)

    def update(self, preds, target):
        p = preds.contiguous()
        t = target.contiguous()
        if p.dtype not in (torch.uint8, torch.int32, torch.int64):
            p = p.long()
        if t.dtype not in (torch.uint8, torch.int32, torch.int64):
            t = t.long()
        ops.confmat_update(self.confmat, t, p)

    def _reduce(self, cm):
        per, w, m = ops.jaccard(cm)
        if self.average in (None, "none"):
            return per
        return w if self.average == "weighted" else m

    def compute(self):
        cm = self.confmat
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            cm = cm.clone()
            torch.distributed.all_reduce(cm)  # dist_reduce_fx="sum" 
        return self._reduce(cm)

    def forward(self, preds, target):
        batch = torch.zeros_like(self.confmat)
        p, t = preds.contiguous(), target.contiguous()
        if p.dtype not in (torch.uint8, torch.int32, torch.int64):
            p = p.long()
        if t.dtype not in (torch.uint8, torch.int32, torch.int64):
            t = t.long()
        ops.confmat_update(batch, t, p)
        self.confmat += batch
        return self._reduce(batch)

    def reset(self):
        self.confmat.zero_()


class MeanMetric(nn.Module):
    def __init__(self, **_):
        super().__init__()
        self.register_buffer("total", torch.zeros((), dtype=torch.float32), persistent=False)   # see MulticlassJaccardIndex
        self.register_buffer("weight", torch.zeros((), dtype=torch.float32), persistent=False)

    def update(self, value, weight=1.0):
        v = value.detach().float() if torch.is_tensor(value) else torch.tensor(float(value), device=self.total.device)
        self.total += v.to(self.total.device).sum() * weight
        self.weight += weight * (v.numel() if torch.is_tensor(v) else 1)

    def compute(self):
        tot, w = self.total, self.weight
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            buf = torch.stack([tot, w])
            torch.distributed.all_reduce(buf)
            tot, w = buf[0], buf[1]
        return tot / w

    def reset(self):
        self.total.zero_()
        self.weight.zero_()
