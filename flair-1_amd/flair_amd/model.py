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


def _ptr3(ts):
    import ctypes as C
    return (C.c_void_p * 3)(*[None if t is None else L.ptr(t) for t in ts])


class _MlpFn(torch.autograd.Function):
    """The three Linear + Dropout + ReLU layers as ONE HIP kernel each way (flair_metadata_mlp_forward / _backward)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, masks):
        x = x.detach().float().contiguous()
        B = x.shape[0]
        ws = [w.detach().float().contiguous() for w in (w1, w2, w3)]
        bs = [b.detach().float().contiguous() for b in (b1, b2, b3)]
        h1 = torch.empty(B, 64, dtype=torch.float32, device=x.device)
        h2 = torch.empty(B, 32, dtype=torch.float32, device=x.device)
        out = torch.empty(B, 16, dtype=torch.float32, device=x.device)
        L.check(L.lib().flair_metadata_mlp_forward(L.ptr(x), _ptr3(ws), _ptr3(bs), None if masks is None else _ptr3(masks),
                                                   L.ptr(h1), L.ptr(h2), L.ptr(out), B, L.stream()), "metadata_mlp_forward")
        ctx.save_for_backward(x, h1, h2, out, *ws)
        ctx.masks = masks
        return out

    @staticmethod
    def backward(ctx, g):
        x, h1, h2, out, w1, w2, w3 = ctx.saved_tensors
        B = x.shape[0]
        dw = [torch.empty_like(w) for w in (w1, w2, w3)]
        db = [torch.empty(n, dtype=torch.float32, device=x.device) for n in (64, 32, 16)]
        scratch = torch.empty(B * 112, dtype=torch.float32, device=x.device)
        L.check(L.lib().flair_metadata_mlp_backward(L.ptr(x), L.ptr(h1), L.ptr(h2), L.ptr(out), _ptr3([w1, w2, w3]),
                                                    None if ctx.masks is None else _ptr3(ctx.masks),
                                                    L.ptr(g.detach().float().contiguous()), B, _ptr3(dw), _ptr3(db),
                                                    L.ptr(scratch), L.stream()), "metadata_mlp_backward")
        return None, dw[0], db[0], dw[1], db[1], dw[2], db[2], None


class MetadataMLP(nn.Module):
    """Light MLP to encode metadata (model.py:74-96): 45 -> 64 -> 32 -> 16, Dropout(0.4) + ReLU each.  Same module tree
    and state_dict keys (enc_mlp.0 / .3 / .6) as the reference; on a HIP device the forward and backward are one kernel
    each (csrc/metadata_mlp.hip), with the dropout masks drawn here from torch's generator in training mode.  Host
    tensors (the CPU-side unit tests of the mirror) take the plain nn.Sequential."""

    def __init__(self):
        super().__init__()
        self.enc_mlp = nn.Sequential(
            nn.Linear(45, 64), nn.Dropout(0.4), nn.ReLU(),
            nn.Linear(64, 32), nn.Dropout(0.4), nn.ReLU(),
            nn.Linear(32, 16), nn.Dropout(0.4), nn.ReLU(),
        )

    def dropout_masks(self, B, device):
        """Bernoulli(keep = 0.6) / 0.6 per layer, as nn.Dropout(0.4) applies them; None in eval mode."""
        if not self.training:
            return None
        return [torch.bernoulli(torch.full((B, n), 0.6, device=device)) / 0.6 for n in (64, 32, 16)]

    def forward(self, x, masks="auto"):
        if not x.is_cuda:
            return self.enc_mlp(x)
        if x.dim() != 2 or x.shape[1] != 45:
            raise ValueError("MetadataMLP expects (B, 45) metadata vectors")
        if masks == "auto":
            masks = self.dropout_masks(x.shape[0], x.device)
        if x.shape[0] > 256:   # the backward kernel takes up to 256 samples per launch: larger batches go in slices
            parts = [self.forward(x[i:i + 256], None if masks is None else [m[i:i + 256] for m in masks])
                     for i in range(0, x.shape[0], 256)]
            return torch.cat(parts, 0)
        l1, l2, l3 = self.enc_mlp[0], self.enc_mlp[3], self.enc_mlp[6]
        return _MlpFn.apply(x, l1.weight, l1.bias, l2.weight, l2.bias, l3.weight, l3.bias, masks)


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
            # model.py:43-50: AutoModelForSemanticSegmentation.from_pretrained(org_model, num_labels=...).  No hub here: the
            # SegFormer family (BASELINE config 5) is built natively from its published geometry, weights arrive through
            # load_state_dict; anything else raises.  Inference only.
            from .segformer import SegformerForSemanticSegmentation, config_for
            hf = config["model_framework"]["HuggingFace"]
            kw = config_for(hf["org_model"])
            kw.update({k: hf[k] for k in ("depths", "decoder_hidden_size") if k in hf})
            self.seg_model = SegformerForSemanticSegmentation(num_channels=n_channels, num_labels=n_classes,
                                                              compute_dtype=compute_dtype, **kw)

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
        elif self.model_provider == "HuggingFace":
            output = self.seg_model(x).logits   # model.py:66-68
        else:
            output = self.seg_model(x)
        return output
