"""Host-side mirror of ``transformers.SegformerForSemanticSegmentation`` (inference) over the HIP executor.

The reference's zone_detect builds its HuggingFace model with ``AutoModelForSemanticSegmentation.from_pretrained``
(/root/reference/src/zone_detect/model.py:42-50, src/flair/model.py:43-50) and runs ``model(imgs).logits``
(compare.py:31-36, model.py:66-68); BASELINE config 5 names SegFormer-MiT-B2 with 5 input channels.  This module is that
object by interface: ``forward(pixel_values) -> output.logits`` of shape (B, num_labels, H/4, W/4), ``state_dict()`` /
``load_state_dict()`` with the library's key names and shapes (transformers 5.x: ``segformer.stages.<i>...``,
``decode_head...``), ``eval()`` only.  ``forward_full`` additionally returns the logits after the x4 bilinear upsample
(align_corners=False) that softmax / margin crop / convert need at tile resolution (the library's own loss path does the
same interpolation).

The nn.Module tree below holds PARAMETER CONTAINERS only; all arithmetic runs in libflair_hip.so
(flair_segformer_forward), host tensors are refused.  Every tensor is a view of one flat fp32 device buffer laid out by the
native tensor table.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import re
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import _lib as L

# published MiT geometries (SegFormer paper, table 6): depths per stage; hidden sizes 64/128/320/512, heads 1/2/5/8 and
# reduction ratios 8/4/2/1 are common to B1 .. B5; decode-head width 256 for B1, 768 from B2 on
MIT_DEPTHS = {"b1": (2, 2, 2, 2), "b2": (3, 4, 6, 3), "b3": (3, 4, 18, 3), "b4": (3, 8, 27, 3), "b5": (3, 6, 40, 3)}


def config_for(org_model: str) -> dict:
    """'nvidia/mit-b2', 'nvidia/segformer-b2-finetuned-...' -> constructor keywords (no hub access here: geometry only)."""
    m = re.search(r"(?:mit|segformer)[-_]?(b[1-5])", org_model.lower())
    if not m:
        raise NotImplementedError(f"HuggingFace model {org_model!r}: only SegFormer / MiT-B1..B5 are built natively")
    v = m.group(1)
    return {"depths": MIT_DEPTHS[v], "decoder_hidden_size": 256 if v == "b1" else 768}


class _Box(nn.Module):
    """A node of the parameter tree (children are added by dotted name)."""

    def forward(self, *a, **k):
        raise RuntimeError("flair_amd: sub-modules are parameter containers; call the model")


class SegformerForSemanticSegmentation(nn.Module):
    def __init__(self, num_channels=3, num_labels=150, depths=(3, 4, 6, 3), hidden_sizes=(64, 128, 320, 512),
                 num_attention_heads=(1, 2, 5, 8), sr_ratios=(8, 4, 2, 1), decoder_hidden_size=768, compute_dtype=None,
                 initializer_range=0.02):
        super().__init__()
        self._dt = L.dtype_code(compute_dtype if compute_dtype is not None else os.environ.get("FLAIR_AMD_DTYPE", "f32"))
        self.num_channels, self.num_labels = int(num_channels), int(num_labels)
        self.config = SimpleNamespace(num_channels=self.num_channels, num_labels=self.num_labels, depths=tuple(depths),
                                      hidden_sizes=tuple(hidden_sizes), num_attention_heads=tuple(num_attention_heads),
                                      sr_ratios=tuple(sr_ratios), decoder_hidden_size=int(decoder_hidden_size))
        arr = lambda v: (C.c_int * 4)(*[int(x) for x in v])
        h = C.c_void_p()
        L.check(L.lib().flair_segformer_create(C.byref(h), self.num_channels, self.num_labels, arr(depths), arr(hidden_sizes),
                                               arr(num_attention_heads), arr(sr_ratios), int(decoder_hidden_size), self._dt),
                "flair_segformer_create")
        object.__setattr__(self, "_h", h)
        self._layout = self._query_layout()
        self._n = L.lib().flair_segformer_param_count(h)
        g = torch.Generator().manual_seed(torch.initial_seed() & 0x7FFFFFFF)
        for name, (shape, off, kind) in self._layout.items():
            leaf = name.rsplit(".", 1)[1]
            if kind == 1:
                t = torch.zeros(shape) if leaf == "running_mean" else torch.ones(shape)
            elif leaf == "bias":
                t = torch.zeros(shape)
            elif len(shape) == 1:            # LayerNorm / BatchNorm weight
                t = torch.ones(shape)
            else:                             # Linear / Conv2d weight: normal(0, initializer_range) like the library
                t = torch.empty(shape).normal_(0.0, initializer_range, generator=g)
            self._attach(name, t, kind)
        self._attach("decode_head.batch_norm.num_batches_tracked", torch.zeros((), dtype=torch.int64), 1)
        self._flat = None
        self._version = -1
        self._ws = None
        self.eval()

    # ---- parameter tree
    def _attach(self, name, tensor, kind):
        mod = self
        *path, leaf = name.split(".")
        for p in path:
            if p not in mod._modules:
                mod.add_module(p, _Box())
            mod = mod._modules[p]
        if kind == 0:
            mod.register_parameter(leaf, nn.Parameter(tensor, requires_grad=False))
        else:
            mod.register_buffer(leaf, tensor)

    def _tensor(self, name):
        mod = self
        *path, leaf = name.split(".")
        for p in path:
            mod = mod._modules[p]
        return mod, leaf

    def _query_layout(self):
        l = L.lib()
        out = {}
        name = C.create_string_buffer(160)
        shape = (C.c_int64 * 4)()
        nd, kind, off = C.c_int(), C.c_int(), C.c_int64()
        for i in range(l.flair_segformer_num_tensors(self._h)):
            L.check(l.flair_segformer_tensor_info(self._h, i, name, 160, shape, C.byref(nd), C.byref(off), C.byref(kind)))
            out[name.value.decode()] = (tuple(shape[d] for d in range(nd.value)), off.value, kind.value)
        return out

    def train(self, mode=True):
        if mode:
            raise RuntimeError("flair_amd.SegformerForSemanticSegmentation is inference-only (zone_detect never trains)")
        return super().train(False)

    def _flatten(self):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise L.FlairHipError("flair_amd.SegformerForSemanticSegmentation runs on a HIP device only: call .cuda() first")
        ok = self._flat is not None and self._flat.device == dev
        if ok:
            base = self._flat.data_ptr()
            version = 0
            for name, (shape, off, _) in self._layout.items():
                mod, leaf = self._tensor(name)
                t = getattr(mod, leaf)
                if t.data_ptr() != base + 4 * off or t.dtype != torch.float32:
                    ok = False
                    break
                version += t._version
        if ok:
            if version != self._version:      # an in-place update (load_state_dict, copy_, ...) since the last forward:
                self.weights_changed()        # the library re-packs its cached weight layouts
                self._version = version
            return self._flat
        flat = torch.zeros(self._n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for name, (shape, off, _) in self._layout.items():
                mod, leaf = self._tensor(name)
                t = getattr(mod, leaf)
                view = flat[off:off + math.prod(shape)].view(shape)
                view.copy_(t.detach().to(device=dev, dtype=torch.float32))
                t.data = view
        self._flat = flat
        self._version = sum(getattr(*self._tensor(name))._version for name in self._layout)
        self.weights_changed()
        return flat

    def weights_changed(self):
        """The library keeps packed copies of the weights between forwards; in-place updates through the module's parameters are
        noticed (tensor version counters), updates through ``.data`` or raw pointers are not — call this after such an update."""
        L.lib().flair_segformer_weights_changed(self._h)

    # ---- native forward
    def _run(self, x, want_quarter, want_full):
        if not x.is_cuda:
            raise L.FlairHipError("flair_amd.SegformerForSemanticSegmentation needs HIP tensors (no CPU fallback)")
        flat = self._flatten()
        if x.device != flat.device:
            raise L.FlairHipError(f"tensor on {x.device} passed to a model on {flat.device}")
        x = x.detach().to(torch.float32).contiguous()
        if x.dim() != 4 or x.shape[1] != self.num_channels:
            raise RuntimeError(f"expected input (B,{self.num_channels},H,W), got {tuple(x.shape)}")
        B, _, H, W = x.shape
        l = L.lib()
        with torch.cuda.device(flat.device):
            need = l.flair_segformer_workspace_bytes(self._h, B, H, W)
            if need < 0:
                raise RuntimeError(f"unsupported tile size {H}x{W}: H and W must be multiples of 32 with (H/32)*(W/32) a "
                                   "multiple of 16 and at most 256 (e.g. 128, 256, 512)")
            if self._ws is None or self._ws.numel() < need or self._ws.device != flat.device:
                self._ws = torch.empty(need, dtype=torch.uint8, device=flat.device)
            lq = torch.empty(B, self.num_labels, H // 4, W // 4, dtype=torch.float32, device=x.device) if want_quarter else None
            lf = torch.empty(B, self.num_labels, H, W, dtype=torch.float32, device=x.device) if want_full else None
            L.check(l.flair_segformer_forward(self._h, L.ptr(flat), L.ptr(x), L.ptr(lq), L.ptr(lf), B, H, W, L.ptr(self._ws),
                                              self._ws.numel(), L.stream()), "flair_segformer_forward")
        return lq, lf

    @torch.no_grad()
    def forward(self, pixel_values, labels=None, **_):
        if labels is not None:
            raise RuntimeError("inference-only: no loss")
        lq, _ = self._run(pixel_values, True, False)
        return SimpleNamespace(logits=lq, loss=None)

    @torch.no_grad()
    def forward_full(self, pixel_values):
        """logits after nn.functional.interpolate(size=input size, mode='bilinear', align_corners=False): (B, labels, H, W)"""
        return self._run(pixel_values, False, True)[1]

    def __del__(self):
        try:
            h = self.__dict__.get("_h")
            if h:
                L.lib().flair_segformer_destroy(h)
        except Exception:  # noqa: BLE001
            pass
