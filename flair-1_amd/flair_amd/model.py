"""Mirror of /root/reference/src/flair/model.py (``FLAIR_ModelFactory`` :7-70, ``MetadataMLP`` :74-96).

Same constructor argument (the YAML config dict), same attributes (``seg_model``, ``enc``) and the same
``forward(x, met=None)``.  Differences, all documented in DESIGN.md:
  * ``smp.create_model`` is served by ``flair_amd.unet.create_model`` (the HIP executor);
  * the reference's ``NameError`` at model.py:32 (bare ``model_provider``) is fixed to ``self.model_provider``;
  * the metadata broadcast ``x_enc.unsqueeze(1).unsqueeze(-1).repeat(1,512,1,16)`` + ``torch.add``
    (model.py:59-60; element [b,c,h,w] += x_enc[b,h]) runs as one in-place HIP kernel on a copy of feats[-1]
    instead of materialising the (B,512,16,16) repeat;
  * the HuggingFace provider branch (model.py:43-50,66-68) is outside the hot path and raises.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib as L
from .unet import create_model


class _AddRowVec(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, x_enc):
        out = feat.detach().clone().contiguous()
        N, C, H, W = out.shape
        L.check(L.lib().flair_add_rowvec_nchw(L.ptr(out), L.ptr(x_enc.detach().float().contiguous()), N, C, H, W, L.stream()))
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g.sum(dim=(1, 3))


class MetadataMLP(nn.Module):
    """Light MLP to encode metadata (model.py:74-96): 45 -> 64 -> 32 -> 16, Dropout(0.4) + ReLU each."""

    def __init__(self):
        super().__init__()
        self.enc_mlp = nn.Sequential(
            nn.Linear(45, 64), nn.Dropout(0.4), nn.ReLU(),
            nn.Linear(64, 32), nn.Dropout(0.4), nn.ReLU(),
            nn.Linear(32, 16), nn.Dropout(0.4), nn.ReLU(),
        )

    def forward(self, x):
        return self.enc_mlp(x)


class FLAIR_ModelFactory(nn.Module):
    def __init__(self, config, compute_dtype=None):
        super().__init__()
        self.model_provider = config["model_framework"]["model_provider"]
        self.use_metadata = config["use_metadata"]
        n_channels = int(len(config["channels"]))
        n_classes = int(len(config["classes"]))
        if self.use_metadata and self.model_provider == "SegmentationModelsPytorch":
            self.enc = MetadataMLP()
        if self.model_provider == "SegmentationModelsPytorch":
            encoder, architecture = config["model_framework"]["SegmentationModelsPytorch"]["encoder_decoder"].split("_")
            kw = {} if compute_dtype is None else {"compute_dtype": compute_dtype}
            if "encoder_weights" in config.get("model_framework", {}).get("SegmentationModelsPytorch", {}):
                kw["encoder_weights"] = config["model_framework"]["SegmentationModelsPytorch"]["encoder_weights"]
            self.seg_model = create_model(arch=architecture, encoder_name=encoder, classes=n_classes,
                                          in_channels=n_channels, **kw)
        elif self.model_provider == "HuggingFace":
            raise NotImplementedError("flair_amd covers the SegmentationModelsPytorch U-Net/ResNet34 hot path only "
                                      "(HuggingFace models are out of scope, SURVEY.md §8f row f3)")

    def forward(self, x, met=None):
        if self.use_metadata == True and self.model_provider == "SegmentationModelsPytorch":  # noqa: E712
            feats = self.seg_model.encoder(x)
            x_enc = self.enc(met)
            if feats[-1].shape[1:] != (512, 16, 16):
                # the reference hard-codes repeat(1,512,1,16): only 512x512 tiles are valid (quirk Q4)
                raise RuntimeError("metadata fusion expects a (B,512,16,16) bottleneck (512x512 input tiles)")
            feats[-1] = _AddRowVec.apply(feats[-1], x_enc)
            output = self.seg_model.decoder(*feats)
            output = self.seg_model.segmentation_head(output)
        else:
            output = self.seg_model(x)
        return output
