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


class _Phase:
    """Running loss and Jaccard state of one phase ('train' / 'val') of an epoch."""

    def __init__(self, num_classes, device, per_class):
        self.loss = MeanMetric().to(device)
        self.miou = MulticlassJaccardIndex(num_classes=num_classes, average="weighted").to(device)
        self.iou = MulticlassJaccardIndex(num_classes=num_classes, average=None).to(device) if per_class else None

    def update(self, loss, preds, targets):
        self.loss.update(loss)
        self.miou(preds=preds, target=targets)
        if self.iou is not None:
            self.iou(preds=preds, target=targets)

    def finish(self):
        out = (self.loss.compute(), self.miou.compute(), None if self.iou is None else self.iou.compute())
        for m in (self.loss, self.miou, self.iou):
            if m is not None:
                m.reset()
        return out


# attribute names of the reference's module (task_module.py:31-51) -> (phase, member of _Phase)
_LEGACY = {"train_loss": ("train", "loss"), "train_metrics": ("train", "miou"), "val_loss": ("val", "loss"),
           "val_metrics": ("val", "miou"), "val_iou": ("val", "iou")}
_LOG_KW = dict(on_step=False, on_epoch=True, logger=True, rank_zero_only=True, sync_dist=True)


class segmentation_task_training(_Base):
    """The reference's LightningModule (task_module.py:8-168) by interface: constructor arguments, hook names, logged keys
    (train_loss, val_loss, val_miou, val_iou_<class> for classes of non-zero weight) and the ``*_epoch_*`` attributes Lightning
    callbacks read.  ``step`` is the fused kernel launch; the epoch bookkeeping is one small state object per phase."""

    def __init__(self, model, class_infos: dict, criterion=None, optimizer=None, use_metadata: bool = False,
                 scheduler: bool = None):
        super().__init__()
        self.model, self.criterion, self.optimizer, self.scheduler = model, criterion, optimizer, scheduler
        self.use_metadata = use_metadata
        self.num_classes = len(class_infos)
        self.class_names = [v[1] for v in class_infos.values()]
        self.class_weigths = [v[0] for v in class_infos.values()]   # (the reference's spelling)
        object.__setattr__(self, "_phases", {})

    def setup(self, stage=None):
        dev = next(self.model.parameters()).device
        wanted = {"fit": ("train", "val"), "validate": ("val",)}.get(stage, ())
        for ph in wanted:
            # 'fit' tracks the per-class IoU of the validation phase as well (task_module.py:43-46)
            self._phases[ph] = _Phase(self.num_classes, dev, per_class=(ph == "val" and stage == "fit"))
            setattr(self, ph + "_epoch_loss", None)
            setattr(self, ph + "_epoch_metrics", None)

    def __getattr__(self, name):
        if name in _LEGACY:
            ph, member = _LEGACY[name]
            phases = self.__dict__.get("_phases", {})
            if ph in phases:
                return getattr(phases[ph], member)
        return super().__getattr__(name)

    def forward(self, input_im, input_met):
        return self.model(input_im, input_met)

    def step(self, batch):
        meta = batch["mtd"] if self.use_metadata else ""
        return fused_step(self.forward(batch["img"], meta), batch["msk"], self.criterion)

    def _run(self, phase, batch):
        loss, preds, targets = self.step(batch)
        self._phases[phase].update(loss, preds, targets)
        return loss

    def training_step(self, batch, batch_idx):
        return self._run("train", batch)

    def validation_step(self, batch, batch_idx):
        return self._run("val", batch)

    def _log(self, name, value, prog_bar=True):
        if _Base is not nn.Module:   # a real LightningModule
            self.log(name, value, prog_bar=prog_bar, **_LOG_KW)

    def _finish(self, phase):
        loss, miou, per_class = self._phases[phase].finish()
        setattr(self, phase + "_epoch_loss", loss)
        setattr(self, phase + "_epoch_metrics", miou)
        self._log(phase + "_loss", loss)
        return miou, per_class

    def on_train_epoch_end(self):
        self._finish("train")

    def on_validation_epoch_end(self):
        miou, per_class = self._finish("val")
        self._log("val_miou", miou)
        self.val_iou_per_class = {}
        if per_class is not None:
            for name, weight, iou in zip(self.class_names, self.class_weigths, per_class):
                if weight != 0:
                    self.val_iou_per_class[name] = iou
                    self._log(f"val_iou_{name}", iou.item(), prog_bar=False)

    def configure_optimizers(self):
        if self.scheduler is None:
            return self.optimizer
        return {"optimizer": self.optimizer,
                "lr_scheduler": {"scheduler": self.scheduler, "interval": "epoch", "monitor": "val_loss", "frequency": 1,
                                 "strict": True, "name": "Scheduler"}}


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
