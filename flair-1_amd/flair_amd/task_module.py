"""Mirror of /root/reference/src/flair/task_module.py over the fused HIP head.

Same class names, constructor arguments, hook names and return values as the reference
(``segmentation_task_training`` task_module.py:8-168, ``segmentation_task_predict`` :174-213), so
``src/flair/tasks_utils.get_segmentation_module`` can build these instead by changing one import.
What changes underneath: ``step()`` makes ONE kernel launch for
one-hot->label argmax + weighted CE + argmax(softmax) + int32 targets (reference lines 71-79), and the
Jaccard metrics are device-side confusion matrices (``flair_amd.head``) instead of torchmetrics.
If pytorch_lightning is importable the classes derive from ``pl.LightningModule``; otherwise from
``nn.Module`` (the hooks are then driven by ``flair_amd.train`` or by the tests).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .head import MeanMetric, MulticlassJaccardIndex

try:  # pragma: no cover - lightning is absent in the build image
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    _Base = nn.Module


class _StepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, msk, weight):
        loss, dl, preds, targets = ops.ce_head(logits.detach().float().contiguous(), msk.contiguous(), weight,
                                               want_dlogits=True, want_preds="i64", want_targets=True)
        ctx.save_for_backward(dl)
        ctx.mark_non_differentiable(preds, targets)
        return loss, preds, targets

    @staticmethod
    def backward(ctx, g, _gp, _gt):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None


def fused_step(logits, msk, criterion):
    """task_module.py:71-79 in one launch.  msk: fp32 one-hot (B,C,H,W) (reference layout) or (B,H,W) ints."""
    w = getattr(criterion, "weight", None)
    w = None if w is None else w.to(device=logits.device, dtype=torch.float32)
    loss, preds, targets = _StepFn.apply(logits, msk, w)
    return loss, preds.flatten(start_dim=1), targets.flatten(start_dim=1)


class segmentation_task_training(_Base):
    def __init__(self, model, class_infos: dict, criterion=None, optimizer=None, use_metadata: bool = False,
                 scheduler: bool = None):
        super().__init__()
        self.model = model
        self.criterion = criterion
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.use_metadata = use_metadata
        self.num_classes = len(class_infos)
        self.class_names = [class_infos[i][1] for i in class_infos]
        self.class_weigths = [class_infos[i][0] for i in class_infos]

    def setup(self, stage=None):
        dev = next(self.model.parameters()).device
        if stage == "fit":
            self.train_epoch_loss, self.val_epoch_loss = None, None
            self.train_epoch_metrics, self.val_epoch_metrics = None, None
            self.train_metrics = MulticlassJaccardIndex(num_classes=self.num_classes, average="weighted").to(dev)
            self.val_metrics = MulticlassJaccardIndex(num_classes=self.num_classes, average="weighted").to(dev)
            self.val_iou = MulticlassJaccardIndex(num_classes=self.num_classes, average=None).to(dev)
            self.train_loss = MeanMetric().to(dev)
            self.val_loss = MeanMetric().to(dev)
        elif stage == "validate":
            self.val_epoch_loss, self.val_epoch_metrics = None, None
            self.val_metrics = MulticlassJaccardIndex(num_classes=self.num_classes, average="weighted").to(dev)
            self.val_loss = MeanMetric().to(dev)

    def forward(self, input_im, input_met):
        return self.model(input_im, input_met)

    def step(self, batch):
        if self.use_metadata:
            images, metadata, targets = batch["img"], batch["mtd"], batch["msk"]
        else:
            images, metadata, targets = batch["img"], "", batch["msk"]
        logits = self.forward(images, metadata)
        return fused_step(logits, targets, self.criterion)

    def training_step(self, batch, batch_idx):
        loss, preds, targets = self.step(batch)
        self.train_loss.update(loss)
        self.train_metrics(preds=preds, target=targets)
        return loss

    def _log(self, *a, **k):
        if hasattr(super(), "log") and _Base is not nn.Module:
            self.log(*a, **k)

    def on_train_epoch_end(self):
        self.train_epoch_loss = self.train_loss.compute()
        self.train_epoch_metrics = self.train_metrics.compute()
        self._log("train_loss", self.train_epoch_loss, on_step=False, on_epoch=True, prog_bar=True, logger=True,
                  rank_zero_only=True, sync_dist=True)
        self.train_loss.reset()
        self.train_metrics.reset()

    def validation_step(self, batch, batch_idx):
        loss, preds, targets = self.step(batch)
        self.val_loss.update(loss)
        self.val_metrics(preds=preds, target=targets)
        self.val_iou(preds=preds, target=targets)
        return loss

    def on_validation_epoch_end(self):
        self.val_epoch_loss = self.val_loss.compute()
        self.val_epoch_metrics = self.val_metrics.compute()
        iou_per_class = self.val_iou.compute()
        self._log("val_loss", self.val_epoch_loss, on_step=False, on_epoch=True, prog_bar=True, logger=True,
                  rank_zero_only=True, sync_dist=True)
        self._log("val_miou", self.val_epoch_metrics, on_step=False, on_epoch=True, prog_bar=True, logger=True,
                  rank_zero_only=True, sync_dist=True)
        self.val_iou_per_class = {}
        for class_name, class_weight, iou in zip(self.class_names, self.class_weigths, iou_per_class):
            if class_weight == 0:
                continue
            self.val_iou_per_class[class_name] = iou
            self._log(f"val_iou_{class_name}", iou.item(), on_step=False, on_epoch=True, prog_bar=False, logger=True,
                      rank_zero_only=True, sync_dist=True)
        self.val_loss.reset()
        self.val_metrics.reset()
        self.val_iou.reset()

    def configure_optimizers(self):
        if self.scheduler is not None:
            lr_scheduler_config = {"scheduler": self.scheduler, "interval": "epoch", "monitor": "val_loss",
                                   "frequency": 1, "strict": True, "name": "Scheduler"}
            return {"optimizer": self.optimizer, "lr_scheduler": lr_scheduler_config}
        return self.optimizer


class segmentation_task_predict(_Base):
    def __init__(self, model, num_classes: int, use_metadata: bool = False):
        super().__init__()
        self.model = model
        self.num_classes = num_classes
        self.use_metadata = use_metadata

    def forward(self, input_im, input_met):
        return self.model(input_im, input_met)

    def step(self, batch):
        if self.use_metadata == True:  # noqa: E712 (mirrors the reference)
            images, metadata, targets = batch["img"], batch["mtd"], batch["msk"]
        else:
            images, metadata, targets = batch["img"], "", batch["msk"]
        logits = self.forward(images, metadata)
        with torch.no_grad():
            _, _, preds, tg = ops.ce_head(logits.detach().float().contiguous(), targets.contiguous(), None,
                                          want_dlogits=False, want_preds="i64", want_targets=True)
        return preds.flatten(start_dim=1), tg.flatten(start_dim=1)

    def predict_step(self, batch, batch_idx, dataloader_idx=0):
        if self.use_metadata == True:  # noqa: E712
            logits = self.forward(batch["img"], batch["mtd"])
        else:
            logits = self.forward(batch["img"], "")
        batch["preds"] = ops.softmax_argmax(logits.detach().float().contiguous(), want="i64")
        return batch
