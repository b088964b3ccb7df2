"""Checkpoint interop: the reference's loaders over this package's modules.

  * ``load_checkpoint(conf, seg_module, exit_on_fail)``  src/flair/main.py:76-146 — ``.ckpt`` (Lightning, key
    ``state_dict``) or ``.pth``/``.pt`` (bare state dict); when the checkpoint's class count differs, the tensors of
    the segmentation head and of ``criterion.weight`` whose shapes mismatch are replaced (zeros of the first
    ``num_classes`` rows / the config's class weights) and the rest loads with ``strict=False``.
  * ``get_module(checkpoint)`` / ``load_model(config)``    src/zone_detect/model.py:61-88 — strips the
    ``model.seg_model.`` prefix of a Lightning checkpoint and loads strictly into the bare segmentation model.
  * ``save_checkpoint`` / ``resume``                        the subset of a Lightning ``.ckpt`` the reference reads
    back (``state_dict``, optimizer and LR-scheduler state, epoch / global_step — src/flair/tasks.py:47-56,96-101).

Key names and shapes are smp-0.3.3's (OIHW conv weights, BatchNorm 5-tuples), so IGNF's published checkpoints load.
Files are read with ``torch.load(..., weights_only=True)``: nothing from the file is executed.
"""
from __future__ import annotations

import os

import torch

from .model import FLAIR_ModelFactory


def _read(path):
    return torch.load(path, map_location="cpu", weights_only=True)


_BAR = "#" * 63


def _state_dict_of(path):
    """The tensor dict inside a checkpoint file: Lightning ``.ckpt`` keeps it under ``state_dict``, ``.pth`` / ``.pt`` ARE it;
    None for any other extension (main.py:98-106)."""
    ext = os.path.splitext(str(path))[1]
    if ext not in (".ckpt", ".pth", ".pt"):
        return None
    blob = _read(path)
    return blob.get("state_dict", blob) if ext == ".ckpt" else blob


def _fail(message, exit_on_fail):
    print(message)
    if exit_on_fail:
        raise SystemExit()


def load_checkpoint(conf, seg_module, exit_on_fail=False):
    """main.py:76-146: load the weights; when the file was trained for another class count, the head / criterion tensors whose
    shapes differ are replaced (criterion.weight <- the config's class weights, head tensors <- zeros of the first
    ``num_classes`` rows) and everything else loads non-strictly.  Same console messages as the reference."""
    path = conf["paths"]["ckpt_model_path"]
    want = len(conf["classes"])
    print("\n" + _BAR)
    try:
        if not (path and os.path.isfile(path)):
            return _fail("Invalid checkpoint file path.", exit_on_fail)
        sd = _state_dict_of(path)
        if sd is None:
            return _fail("Invalid file extension.", exit_on_fail)
        have = next((v.shape[0] for k, v in sd.items() if "classifier.weight" in k or "criterion.weight" in k), None)
        if have == want:
            print("--------------- Loaded model weights from checkpoint with matching number of classes. ---------------")
        else:
            print(f"Number of classes in checkpoint ({have}) does not match the current number of classes ({want}). "
                  "Proceeding with modifications.")
            own = seg_module.state_dict()
            for k in [k for k, v in sd.items() if k in own and v.shape != own[k].shape and ("head" in k or "criterion" in k)]:
                old_shape = sd[k].shape
                sd[k] = (torch.FloatTensor([c[0] for c in conf["classes"].values()]) if "criterion" in k
                         else torch.zeros_like(sd[k][:want]))
                print("-", k, "has been modified.")
                print(old_shape, "  ->  ", sd[k].shape)
        seg_module.load_state_dict(sd, strict=False)
    finally:
        print(_BAR + "\n")


def get_module(checkpoint):
    """zone_detect/model.py:61-79: the bare segmentation model's tensors out of a ``.ckpt`` / ``.pth`` — keys of a Lightning
    module (``model.seg_model.<name>``) lose that prefix, everything else of such a file (criterion, metadata MLP) is dropped."""
    if checkpoint is None or not os.path.isfile(checkpoint):
        print('Error with checkpoint provided: either a .ckpt with a "state_dict" key or an OrderedDict pt/pth file')
        return {}
    weights = _read(checkpoint)
    if str(checkpoint).endswith(".ckpt"):
        weights = weights["state_dict"]
    prefix = "model.seg_model."
    if prefix[:-1] in next(iter(weights)):
        weights = {k.partition(prefix)[2]: v for k, v in weights.items() if k.partition(prefix)[2]}
    return weights


def load_model(config: dict, compute_dtype=None):
    kw = {} if compute_dtype is None else {"compute_dtype": compute_dtype}
    model = FLAIR_ModelFactory(config, **kw).seg_model
    model.load_state_dict(state_dict=get_module(checkpoint=config["model_weights"]), strict=True)
    return model


def save_checkpoint(path, seg_module, optimizer=None, scheduler=None, epoch: int = 0, global_step: int = 0):
    """A ``.ckpt`` with Lightning's top-level keys (tensors moved to the host); ``.pth``/``.pt`` -> bare state dict."""
    sd = {k: v.detach().cpu() for k, v in seg_module.state_dict().items()}
    if str(path).endswith(".ckpt"):
        ckpt = {"epoch": int(epoch), "global_step": int(global_step), "state_dict": sd,
                "optimizer_states": [optimizer.state_dict()] if optimizer is not None else [],
                "lr_schedulers": [scheduler.state_dict()] if scheduler is not None else []}
        torch.save(ckpt, path)
    else:
        torch.save(sd, path)


def resume(path, seg_module, optimizer=None, scheduler=None):
    """Restore what ``Trainer.fit(ckpt_path=...)`` restores for this path; returns (epoch, global_step)."""
    ckpt = _read(path)
    seg_module.load_state_dict(ckpt["state_dict"], strict=True)
    if optimizer is not None and ckpt.get("optimizer_states"):
        optimizer.load_state_dict(ckpt["optimizer_states"][0])
    if scheduler is not None and ckpt.get("lr_schedulers"):
        scheduler.load_state_dict(ckpt["lr_schedulers"][0])
    return int(ckpt.get("epoch", 0)), int(ckpt.get("global_step", 0))
