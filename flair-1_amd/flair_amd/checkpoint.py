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


def load_checkpoint(conf, seg_module, exit_on_fail=False):
    print()
    print("###############################################################")
    ckpt_file_path = conf["paths"]["ckpt_model_path"]
    num_classes = len(conf["classes"])
    if ckpt_file_path and os.path.isfile(ckpt_file_path):
        checkpoint = _read(ckpt_file_path)
        if ckpt_file_path.endswith(".ckpt"):
            state_dict = checkpoint.get("state_dict", checkpoint)
        elif ckpt_file_path.endswith(".pth") or ckpt_file_path.endswith(".pt"):
            state_dict = checkpoint
        else:
            print("Invalid file extension.")
            if exit_on_fail:
                raise SystemExit()
            return
        ckpt_num_classes = None
        for k, v in state_dict.items():
            if "classifier.weight" in k or "criterion.weight" in k:
                ckpt_num_classes = v.shape[0]
                break
        model_state_dict = seg_module.state_dict()
        if ckpt_num_classes is not None and ckpt_num_classes == num_classes:
            seg_module.load_state_dict(state_dict, strict=False)
            print("--------------- Loaded model weights from checkpoint with matching number of classes. ---------------")
        else:
            print(f"Number of classes in checkpoint ({ckpt_num_classes}) does not match the current number of classes "
                  f"({num_classes}). Proceeding with modifications.")
            ignored_layers = [k for k, v in state_dict.items() if k in model_state_dict and v.shape != model_state_dict[k].shape]
            ignored_layers = [i for i in ignored_layers if any(x in i for x in ["head", "criterion"])]
            for k in ignored_layers:
                print("-", k, "has been modified.")
                print(state_dict[k].shape, "  ->  ", flush=True, end="")
                if "criterion" in k:
                    state_dict[k] = torch.FloatTensor([conf["classes"][i][0] for i in conf["classes"]])
                else:
                    state_dict[k] = 0 * torch.abs(state_dict[k][0:num_classes])
                print(state_dict[k].shape)
            seg_module.load_state_dict(state_dict, strict=False)
        print("###############################################################")
    else:
        print("Invalid checkpoint file path.")
        if exit_on_fail:
            raise SystemExit()
        print("###############################################################")
    print()


def get_module(checkpoint):
    if checkpoint is not None and os.path.isfile(checkpoint):
        weights = _read(checkpoint)
        if str(checkpoint).endswith(".ckpt"):
            weights = weights["state_dict"]
    else:
        print('Error with checkpoint provided: either a .ckpt with a "state_dict" key or an OrderedDict pt/pth file')
        return {}
    if "model.seg_model" in list(weights.keys())[0]:
        weights = {k.partition("model.seg_model.")[2]: v for k, v in weights.items()}
        weights = {k: v for k, v in weights.items() if k != ""}
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
