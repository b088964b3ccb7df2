"""Host-side mirror of ``segmentation_models_pytorch.Unet('resnet34')`` over the HIP executor.

Drop-in for the object ``smp.create_model(arch='unet', encoder_name='resnet34', classes, in_channels)``
returns at /root/reference/src/flair/model.py:37-41 (contract: SURVEY.md §8b):

* ``forward(x) -> logits`` (B,classes,H,W) fp32, H and W divisible by 32 else RuntimeError;
* ``.encoder(x) -> list`` of 6 features, ``.decoder(*feats)``, ``.segmentation_head(t)`` (model.py:57-62);
* ``.parameters()`` are leaf ``nn.Parameter``s (fp32, PyTorch OIHW) that receive ``.grad``;
* ``state_dict()`` / ``load_state_dict()`` use the smp-0.3.3 key names and shapes;
* ``.modules()`` contains real ``nn.BatchNorm2d`` objects (src/flair/tasks.py:25-26 looks for them);
* ``train()`` / ``eval()`` switch BatchNorm between batch and running statistics.

The nn.Conv2d / nn.BatchNorm2d sub-modules are PARAMETER CONTAINERS only: all arithmetic runs in
libflair_hip.so on the current HIP stream.  Host tensors are refused (there is no CPU fallback).
Parameters and BN running statistics live in two flat fp32 device buffers (views per tensor) whose
layout the native library defines (``flair_unet_tensor_info``); stage-contiguous gradients make the
bucketed RCCL all-reduce of ``flair_amd.train`` possible.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import warnings

import torch
import torch.nn as nn

from . import _lib as L


def _on_model_device(fn):
    """Native calls launch on the CURRENT device's stream and the library's internal streams / events are created on the
    current device: run them with the model's device current (a model on cuda:1 while cuda:0 is current — two replicas in
    one process, a forgotten set_device — would otherwise launch on the wrong device's stream), and refuse inputs that
    live elsewhere."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        dev = self._flat_p.device if getattr(self, "_flat_p", None) is not None else None
        if dev is None or dev.type != "cuda":
            return fn(self, *args, **kwargs)
        for a in args:
            ts = a if isinstance(a, (list, tuple)) else (a,)
            for t in ts:
                if torch.is_tensor(t) and t.is_cuda and t.device != dev:
                    raise L.FlairHipError(f"tensor on {t.device} passed to a model on {dev}")
        with torch.cuda.device(dev):
            return fn(self, *args, **kwargs)
    return wrapper


# ------------------------------------------------------------------------------------------------ containers
class _BasicBlock(nn.Module):
    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        self.stride = stride

    def forward(self, x):
        raise RuntimeError("flair_amd: sub-modules are parameter containers; call the Unet / encoder / decoder / head")


def _container_forward(self, *a, **k):
    raise RuntimeError("flair_amd: sub-modules are parameter containers; call the Unet / encoder / decoder / head")


class _Encoder(nn.Module):
    """smp ``ResNetEncoder('resnet34')`` key layout; ``__call__`` runs the HIP encoder stage."""

    def __init__(self, owner, in_channels):
        super().__init__()
        object.__setattr__(self, "_owner", owner)
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inpl = 64
        for li, (planes, n) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3)), start=1):
            blocks = []
            for b in range(n):
                stride = 2 if (b == 0 and li > 1) else 1
                blocks.append(_BasicBlock(inpl, planes, stride, downsample=(b == 0 and li > 1)))
                inpl = planes
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        self._out_channels = (in_channels, 64, 64, 128, 256, 512)
        self._depth = 5
        self._in_channels = in_channels

    @property
    def out_channels(self):
        return self._out_channels

    def forward(self, x):
        return self._owner._encoder_forward(x)


class _DecoderBlock(nn.Module):
    def __init__(self, cin, cskip, cout):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(cin + cskip, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))
        self.attention1 = nn.Identity()
        self.conv2 = nn.Sequential(nn.Conv2d(cout, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))
        self.attention2 = nn.Identity()

    forward = _container_forward


class _Decoder(nn.Module):
    def __init__(self, owner):
        super().__init__()
        object.__setattr__(self, "_owner", owner)
        self.center = nn.Identity()
        cin, cskip, cout = (512, 256, 128, 64, 32), (256, 128, 64, 64, 0), (256, 128, 64, 32, 16)
        self.blocks = nn.ModuleList(_DecoderBlock(a, b, c) for a, b, c in zip(cin, cskip, cout))

    def forward(self, *features):
        return self._owner._decoder_forward(*features)


class _Head(nn.Sequential):
    def __init__(self, owner, classes):
        super().__init__(nn.Conv2d(16, classes, 3, padding=1), nn.Identity(), nn.Identity())
        object.__setattr__(self, "_owner", owner)

    def forward(self, x):
        return self._owner._head_forward(x)


# ------------------------------------------------------------------------------------------------ autograd glue
class _WholeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, x, *params):
        logits = owner._c_forward(x, training=True)
        ctx.owner = owner
        ctx.fwd_id = owner._fwd_id
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        owner = ctx.owner
        owner._check_live(ctx.fwd_id)
        grads = owner._c_backward(dlogits.contiguous())
        return (None, None) + tuple(owner._grad_views(grads))


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, x, *params):
        feats = owner._c_encoder_forward(x, training=True)
        ctx.owner = owner
        ctx.fwd_id = owner._fwd_id
        return tuple(feats)

    @staticmethod
    def backward(ctx, *dfeats):
        owner = ctx.owner
        owner._check_live(ctx.fwd_id)
        owner._c_encoder_backward([d.contiguous() for d in dfeats])
        return (None, None) + tuple(owner._stage_grad_views("encoder"))


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, f1, f2, f3, f4, f5, *params):
        out = owner._c_decoder_forward([f1, f2, f3, f4, f5], training=True)
        ctx.owner = owner
        ctx.fwd_id = owner._fwd_id
        return out

    @staticmethod
    def backward(ctx, dout):
        owner = ctx.owner
        owner._check_live(ctx.fwd_id)
        dfeats = owner._c_decoder_backward(dout.contiguous())
        return (None,) + tuple(dfeats) + tuple(owner._stage_grad_views("decoder"))


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, x, *params):
        logits = owner._c_head_forward(x, training=True)
        ctx.owner = owner
        ctx.fwd_id = owner._fwd_id
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        owner = ctx.owner
        owner._check_live(ctx.fwd_id)
        dx = owner._c_head_backward(dlogits.contiguous())
        return (None, dx) + tuple(owner._stage_grad_views("segmentation_head"))


# ------------------------------------------------------------------------------------------------ the model
class Unet(nn.Module):
    """MI355X-native ``smp.Unet(encoder_name='resnet34')``.

    Extra keyword (not in smp): ``compute_dtype`` = 'f32' (parity mode: exact-fp32 MFMA, the
    reference trains in fp32, src/flair/tasks.py:83-93) or 'bf16' (throughput mode).  Default comes
    from the environment variable FLAIR_AMD_DTYPE, else 'f32'.
    """

    def __init__(self, encoder_name="resnet34", encoder_depth=5, encoder_weights="imagenet", decoder_use_batchnorm=True,
                 decoder_channels=(256, 128, 64, 32, 16), decoder_attention_type=None, in_channels=3, classes=1,
                 activation=None, aux_params=None, compute_dtype=None):
        super().__init__()
        if encoder_name != "resnet34":
            raise KeyError(f"Wrong encoder name `{encoder_name}`, supported encoders: ['resnet34']")
        if (encoder_depth != 5 or not decoder_use_batchnorm or tuple(decoder_channels) != (256, 128, 64, 32, 16)
                or decoder_attention_type is not None or activation is not None or aux_params is not None):
            raise ValueError("flair_amd.Unet implements the configuration the reference uses: smp defaults with resnet34")
        self._dt = L.dtype_code(compute_dtype if compute_dtype is not None else os.environ.get("FLAIR_AMD_DTYPE", "f32"))
        self.in_channels, self.classes = int(in_channels), int(classes)
        self.encoder = _Encoder(self, self.in_channels)
        self.decoder = _Decoder(self)
        self.segmentation_head = _Head(self, self.classes)
        self.classification_head = None
        self.name = "u-resnet34"
        self._init_weights(encoder_weights)
        # native handle + layout
        h = C.c_void_p()
        L.check(L.lib().flair_unet_create(C.byref(h), self.in_channels, self.classes, self._dt), "flair_unet_create")
        object.__setattr__(self, "_h", h)
        self._layout = self._query_layout()
        sd_tensors = dict(self.named_parameters())
        sd_tensors.update(dict(self.named_buffers()))
        for name, (shape, off, kind, stage) in self._layout.items():
            t = sd_tensors.get(name)
            if t is None or tuple(t.shape) != shape:
                raise RuntimeError(f"layout mismatch for {name}: native {shape} vs module {None if t is None else tuple(t.shape)}")
        self._param_names = [n for n, (_, _, k, _) in self._layout.items() if k == 0]
        self._buffer_names = [n for n, (_, _, k, _) in self._layout.items() if k == 1]
        self._n_params = L.lib().flair_unet_param_count(h)
        self._n_buffers = L.lib().flair_unet_buffer_count(h)
        self._flat_p = None
        self._flat_b = None
        self._ws = {}
        self._fwd_id = 0
        self._live_id = -1
        self._grads = None

    # -------------------------------------------------------------------------------------------- init
    def _init_weights(self, encoder_weights):
        """smp-0.3.3 initialisation: torchvision ResNet init (encoder), kaiming-uniform decoder, xavier head."""
        first = self.encoder.conv1
        for m in self.encoder.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if self.in_channels != 3:
            first.reset_parameters()  # smp patch_first_conv(pretrained=False)
        for m in self.decoder.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        nn.init.xavier_uniform_(self.segmentation_head[0].weight)
        nn.init.constant_(self.segmentation_head[0].bias, 0)
        if encoder_weights is not None:
            path = os.environ.get("FLAIR_RESNET34_WEIGHTS")
            if path and os.path.isfile(path):
                self.load_encoder_weights(torch.load(path, map_location="cpu", weights_only=True))
            else:
                warnings.warn(
                    f"encoder_weights={encoder_weights!r}: pretrained ResNet34 weights cannot be downloaded here; "
                    "set FLAIR_RESNET34_WEIGHTS=<torchvision resnet34 state_dict .pth> or pass encoder_weights=None. "
                    "Using the seeded random initialisation.")

    def load_encoder_weights(self, sd):
        """torchvision resnet34 state_dict -> encoder, with smp ``patch_first_conv`` for in_channels != 3."""
        sd = {k: v for k, v in sd.items() if not k.startswith("fc.")}
        w = sd["conv1.weight"]
        if self.in_channels == 1:
            sd["conv1.weight"] = w.sum(1, keepdim=True)
        elif self.in_channels != 3:
            nw = torch.empty(w.shape[0], self.in_channels, *w.shape[2:])
            for i in range(self.in_channels):
                nw[:, i] = w[:, i % 3]
            sd["conv1.weight"] = nw * (3 / self.in_channels)
        self.encoder.load_state_dict(sd, strict=True)

    # -------------------------------------------------------------------------------------------- layout
    def _query_layout(self):
        l = L.lib()
        out = {}
        name = C.create_string_buffer(128)
        shape = (C.c_int64 * 4)()
        nd, kind, stage, off = C.c_int(), C.c_int(), C.c_int(), C.c_int64()
        for i in range(l.flair_unet_num_tensors(self._h)):
            L.check(l.flair_unet_tensor_info(self._h, i, name, 128, shape, C.byref(nd), C.byref(off), C.byref(kind), C.byref(stage)))
            out[name.value.decode()] = (tuple(shape[d] for d in range(nd.value)), off.value, kind.value, stage.value)
        return out

    def stage_ranges(self):
        """[(begin, end)] float offsets of the 7 gradient buckets (stem, layer1-4, decoder, head)."""
        b, e = C.c_int64(), C.c_int64()
        res = []
        for s in range(7):
            L.check(L.lib().flair_unet_stage_range(self._h, s, C.byref(b), C.byref(e)))
            res.append((b.value, e.value))
        return res

    def _tensor(self, name):
        mod = self
        *path, leaf = name.split(".")
        for p in path:
            mod = getattr(mod, p) if not p.isdigit() else mod[int(p)]
        return mod, leaf

    def flatten_(self):
        """(Re)bind every parameter / running statistic as a view of the two flat device buffers."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise L.FlairHipError("flair_amd.Unet runs on a HIP device only: call .cuda() first (no CPU fallback)")
        ok = self._flat_p is not None and self._flat_p.device == dev
        if ok:
            base_p, base_b = self._flat_p.data_ptr(), self._flat_b.data_ptr()
            for name in self._param_names + self._buffer_names:
                mod, leaf = self._tensor(name)
                t = getattr(mod, leaf)
                _, off, kind, _ = self._layout[name]
                if t.data_ptr() != (base_p if kind == 0 else base_b) + 4 * off or t.dtype != torch.float32:
                    ok = False
                    break
        if ok:
            return
        flat_p = torch.zeros(self._n_params, dtype=torch.float32, device=dev)
        flat_b = torch.zeros(max(self._n_buffers, 1), dtype=torch.float32, device=dev)
        with torch.no_grad():
            for name in self._param_names + self._buffer_names:
                mod, leaf = self._tensor(name)
                t = getattr(mod, leaf)
                shape, off, kind, _ = self._layout[name]
                n = math.prod(shape)
                view = (flat_p if kind == 0 else flat_b)[off:off + n].view(shape)
                view.copy_(t.detach().to(device=dev, dtype=torch.float32))
                t.data = view
            bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
            flat_n = torch.zeros(len(bns), dtype=torch.int64, device=dev)
            for i, m in enumerate(bns):
                flat_n[i] = m.num_batches_tracked.to(dev)
                m.num_batches_tracked.data = flat_n[i]
        self._flat_p, self._flat_b, self._flat_n = flat_p, flat_b, flat_n

    def flat_parameters(self):
        self.flatten_()
        return self._flat_p

    def flat_buffers(self):
        self.flatten_()
        return self._flat_b

    def _params_list(self, prefix=None):
        names = [n for n in self._param_names if prefix is None or n.startswith(prefix + ".")]
        out = []
        for n in names:
            mod, leaf = self._tensor(n)
            out.append(getattr(mod, leaf))
        return names, out

    def _grad_views(self, grads, prefix=None):
        views = []
        for n in self._param_names:
            if prefix is not None and not n.startswith(prefix + "."):
                continue
            shape, off, _, _ = self._layout[n]
            views.append(grads[off:off + math.prod(shape)].view(shape))
        return views

    def _stage_grad_views(self, prefix):
        return self._grad_views(self._grads, prefix)

    # -------------------------------------------------------------------------------------------- native calls
    def _hh(self, training):
        """The native executor for this call: eval-mode forwards (predict, validation, ZoneDetector) run on a second
        handle, so that one issued between a training forward and its backward — legal in PyTorch — leaves the recorded
        training graph alone."""
        if training:
            return self._h
        h = self.__dict__.get("_h_eval")
        if h is None:
            h = C.c_void_p()
            L.check(L.lib().flair_unet_create(C.byref(h), self.in_channels, self.classes, self._dt), "flair_unet_create")
            object.__setattr__(self, "_h_eval", h)
        return h

    def _workspace(self, B, H, W, training):
        key = "train" if training else "eval"
        plan = self.__dict__.setdefault("_ws_need", {})
        need = plan.get((B, H, W, bool(training)))
        if need is None:   # the dry run walks the whole 47-conv graph on the host: once per shape, not once per step
            need = L.lib().flair_unet_workspace_bytes(self._h, B, H, W, int(training))
            if need < 0:
                raise RuntimeError("flair_unet_workspace_bytes failed")
            plan[(B, H, W, bool(training))] = need
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need or ws.device != self._flat_p.device:
            self._ws[key] = ws = torch.empty(need, dtype=torch.uint8, device=self._flat_p.device)
        return ws

    def check_input_shape(self, x):
        h, w = x.shape[-2:]
        if h % 32 != 0 or w % 32 != 0:
            nh = (h // 32 + 1) * 32 if h % 32 else h
            nw = (w // 32 + 1) * 32 if w % 32 else w
            raise RuntimeError(f"Wrong input shape height={h}, width={w}. Expected image height and width "
                               f"divisible by 32. Consider pad your images to shape ({nh}, {nw}).")

    def _prep(self, x):
        if not x.is_cuda:
            raise L.FlairHipError("flair_amd.Unet needs HIP tensors (no CPU fallback)")
        self.flatten_()
        x = x.detach().to(torch.float32).contiguous()
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise RuntimeError(f"expected input (B,{self.in_channels},H,W), got {tuple(x.shape)}")
        self.check_input_shape(x)
        return x

    def _bump_bn(self, lo=0, hi=None):
        """num_batches_tracked += 1 for BatchNorm layers [lo, hi) (one op on the flat int64 buffer)."""
        self._flat_n[lo:hi] += 1

    # eval forwards of small batches are launch-bound (~150 kernels of a few microseconds: predict_step at the reference's
    # batch size 1, data_module.py:100, ran at 870 tiles/s against 10 000 at batch 32): the third identical call — same weights,
    # shape, workspace and requested outputs — is captured into a HIP graph and replayed from then on (FLAIR_EVAL_GRAPH=0: never)
    _GRAPH_MAX_PIXELS = 4 * 512 * 512

    def _eval_graph(self, gkey, x, want_logits, preds, prob, ws, h):
        ent = self._eval_graphs.get(gkey)
        if ent is None:
            B, _, H, W = x.shape
            ent = {"x": x.clone(), "logits": torch.empty(B, self.classes, H, W, dtype=torch.float32, device=x.device) if want_logits else None,
                   "preds": None if preds is None else torch.empty_like(preds), "prob": None if prob is None else torch.empty_like(prob)}
            g = torch.cuda.CUDAGraph()
            l = L.lib()
            with torch.cuda.graph(g):
                L.check(l.flair_unet_reuse_constants(h, 1), "flair_unet_reuse_constants")
                if preds is not None:
                    L.check(l.flair_unet_want_preds(h, L.ptr(ent["preds"]), L.ptr(ent["prob"])), "flair_unet_want_preds")
                L.check(l.flair_unet_forward(h, L.ptr(self._flat_p), L.ptr(self._flat_b), L.ptr(ent["x"]), L.ptr(ent["logits"]), B, H, W, 0,
                                             L.ptr(ws), ws.numel(), L.stream()), "flair_unet_forward")
            ent["graph"] = g
            self._eval_graphs[gkey] = ent
        ent["x"].copy_(x)       # (capture records, it does not run: the first use replays too)
        ent["graph"].replay()
        if preds is not None:
            preds.copy_(ent["preds"])
            if prob is not None:
                prob.copy_(ent["prob"])
        return None if ent["logits"] is None else ent["logits"].clone()

    @_on_model_device
    def _c_forward(self, x, training, want_logits=True, preds=None, prob=None):
        """want_logits=False (fused trainer): the head's output stays in the workspace as NHWC rows of the compute dtype
        (flair_unet_logits_nhwc) and no fp32 NCHW tensor is produced; returns None.  preds / prob (eval mode, no logits): uint8
        argmax and its fp32 probability from the head convolution's epilogue (flair_unet_want_preds)."""
        x = self._prep(x)
        B, _, H, W = x.shape
        ws = self._workspace(B, H, W, training)
        h = self._hh(training)
        if not training:
            ver = sum(p._version for p in self.parameters()) + sum(b._version for b in self.buffers())
            key = (ver, self._flat_p._version, self._flat_b._version, self.__dict__.get("_native_writes", 0), B, H, W, ws.data_ptr())
            graphs = self.__dict__.setdefault("_eval_graphs", {})
            if self.__dict__.get("_eval_key") != key:
                graphs.clear()
                self._eval_hits = {}
            elif (os.environ.get("FLAIR_EVAL_GRAPH", "1") != "0" and B * H * W <= self._GRAPH_MAX_PIXELS
                  and not torch.cuda.is_current_stream_capturing()):
                gkey = (want_logits, preds is not None, prob is not None)
                hits = self.__dict__.setdefault("_eval_hits", {})
                hits[gkey] = hits.get(gkey, 0) + 1
                if gkey in graphs or hits[gkey] >= 2:   # (the first repeat ran eagerly with reuse: every kernel's one-time set-up is done)
                    return self._eval_graph(gkey, x, want_logits, preds, prob, ws, h)
        logits = torch.empty(B, self.classes, H, W, dtype=torch.float32, device=x.device) if want_logits else None
        if preds is not None:
            L.check(L.lib().flair_unet_want_preds(h, L.ptr(preds), L.ptr(prob)), "flair_unet_want_preds")
        if not training:
            # constant weights between two eval forwards (predict / zone_detect loops): skip the weight pack and the
            # BatchNorm-coefficient launches.  torch bumps a tensor's _version on every in-place write (optimizers,
            # load_state_dict; writes through .data are invisible to it, as they are to autograd); native writers (training forwards: running statistics; the fused
            # trainer's SGD) bump _native_writes.
            # (the parameters / buffers are .data views of the flat buffers: each carries its own version counter)
            ver = sum(p._version for p in self.parameters()) + sum(b._version for b in self.buffers())
            key = (ver, self._flat_p._version, self._flat_b._version, self.__dict__.get("_native_writes", 0), B, H, W, ws.data_ptr())
            if self.__dict__.get("_eval_key") == key:
                L.check(L.lib().flair_unet_reuse_constants(h, 1), "flair_unet_reuse_constants")
            self._eval_key = key
        else:
            self._native_writes = self.__dict__.get("_native_writes", 0) + 1
        L.check(L.lib().flair_unet_forward(h, L.ptr(self._flat_p), L.ptr(self._flat_b), L.ptr(x), L.ptr(logits), B, H, W,
                                           int(training), L.ptr(ws), ws.numel(), L.stream()), "flair_unet_forward")
        if training:
            self._fwd_id += 1
            self._live_id = self._fwd_id
            self._live_ws = ws
            self._bump_bn()
        return logits

    def _check_live(self, fwd_id):
        if fwd_id != self._live_id:
            raise RuntimeError("flair_amd.Unet: the activations of this forward were overwritten by a later training "
                               "forward; call backward before the next training-mode forward")

    def _new_grads(self):
        self._grads = torch.zeros(self._n_params, dtype=torch.float32, device=self._flat_p.device)
        return self._grads

    @_on_model_device
    def _c_backward(self, dlogits=None, dlogits_nhwc=None, grads=None, stage_events=None):
        ws = self._live_ws
        grads = grads if grads is not None else self._new_grads()
        ev = None
        if stage_events is not None:
            ev = (C.c_void_p * 7)(*[e.cuda_event for e in stage_events])
        L.check(L.lib().flair_unet_backward(self._h, L.ptr(self._flat_p), L.ptr(dlogits), L.ptr(dlogits_nhwc), L.ptr(grads),
                                            L.ptr(ws), ws.numel(), L.stream(), ev), "flair_unet_backward")
        return grads

    # ---- split path (model.py:57-62)
    @_on_model_device
    def _c_encoder_forward(self, x, training):
        x = self._prep(x)
        B, _, H, W = x.shape
        ws = self._workspace(B, H, W, training)
        chans = (64, 64, 128, 256, 512)
        feats = [torch.empty(B, c, H >> (i + 1), W >> (i + 1), dtype=torch.float32, device=x.device) for i, c in enumerate(chans)]
        arr = (C.c_void_p * 5)(*[f.data_ptr() for f in feats])
        if not training:
            self._eval_key = None   # the split path re-lays the eval arena: the fused forward's constants are gone
        L.check(L.lib().flair_unet_encoder_forward(self._hh(training), L.ptr(self._flat_p), L.ptr(self._flat_b), L.ptr(x), arr, B, H, W,
                                                   int(training), L.ptr(ws), ws.numel(), L.stream()), "encoder_forward")
        if training:
            self._native_writes = self.__dict__.get("_native_writes", 0) + 1   # running statistics are about to change
        # per mode: an eval-mode split forward must not disturb a training one that still awaits its backward
        self.__dict__.setdefault("_split", {})[bool(training)] = ((B, H, W), ws)
        if training:
            self._split_shape = (B, H, W)
            self._split_ws = ws
            self._fwd_id += 1
            self._live_id = self._fwd_id
            self._live_ws = ws
            self._grads = None
            self._bump_bn(0, 36)
        return feats

    @_on_model_device
    def _c_decoder_forward(self, feats, training):
        (B, H, W), ws = self._split[bool(training)]
        feats = [f.detach().to(torch.float32).contiguous() for f in feats]
        out = torch.empty(B, 16, H, W, dtype=torch.float32, device=feats[0].device)
        arr = (C.c_void_p * 5)(*[f.data_ptr() for f in feats])
        if not training:
            self._eval_key = None
        L.check(L.lib().flair_unet_decoder_forward(self._hh(training), L.ptr(self._flat_p), L.ptr(self._flat_b), arr, L.ptr(out), B, H, W,
                                                   int(training), L.ptr(ws), ws.numel(), L.stream()), "decoder_forward")
        if training:
            self._native_writes = self.__dict__.get("_native_writes", 0) + 1   # decoder running statistics changed
            self._bump_bn(36, 46)
        return out

    @_on_model_device
    def _c_head_forward(self, x, training):
        (B, H, W), ws = self._split[bool(training)]
        x = x.detach().to(torch.float32).contiguous()
        logits = torch.empty(B, self.classes, H, W, dtype=torch.float32, device=x.device)
        if not training:
            self._eval_key = None
        L.check(L.lib().flair_unet_head_forward(self._hh(training), L.ptr(self._flat_p), L.ptr(x), L.ptr(logits), B, H, W, int(training),
                                                L.ptr(ws), ws.numel(), L.stream()), "head_forward")
        return logits

    def _split_grads(self):
        if self._grads is None:
            self._new_grads()
        return self._grads

    @_on_model_device
    def _c_head_backward(self, dlogits):
        B, H, W = self._split_shape
        ws = self._split_ws
        g = self._split_grads()
        dx = torch.empty(B, 16, H, W, dtype=torch.float32, device=dlogits.device)
        L.check(L.lib().flair_unet_head_backward(self._h, L.ptr(self._flat_p), L.ptr(dlogits), L.ptr(dx), L.ptr(g), L.ptr(ws),
                                                 ws.numel(), L.stream()), "head_backward")
        return dx

    @_on_model_device
    def _c_decoder_backward(self, dout):
        B, H, W = self._split_shape
        ws = self._split_ws
        g = self._split_grads()
        chans = (64, 64, 128, 256, 512)
        dfe = [torch.empty(B, c, H >> (i + 1), W >> (i + 1), dtype=torch.float32, device=dout.device) for i, c in enumerate(chans)]
        arr = (C.c_void_p * 5)(*[f.data_ptr() for f in dfe])
        L.check(L.lib().flair_unet_decoder_backward(self._h, L.ptr(self._flat_p), L.ptr(dout), arr, L.ptr(g), L.ptr(ws),
                                                    ws.numel(), L.stream()), "decoder_backward")
        return dfe

    @_on_model_device
    def _c_encoder_backward(self, dfeats):
        ws = self._split_ws
        g = self._split_grads()
        arr = (C.c_void_p * 5)(*[f.data_ptr() for f in dfeats])
        L.check(L.lib().flair_unet_encoder_backward(self._h, L.ptr(self._flat_p), arr, L.ptr(g), L.ptr(ws), ws.numel(),
                                                    L.stream()), "encoder_backward")

    # -------------------------------------------------------------------------------------------- nn.Module surface
    def _needs_graph(self):
        return torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters())

    def forward(self, x):
        if self._needs_graph():
            _, params = self._params_list()
            return _WholeFn.apply(self, x, *params)
        if torch.is_grad_enabled() and not self.training and any(p.requires_grad for p in self.parameters()):
            warnings.warn("flair_amd.Unet: eval-mode forward builds no autograd graph (wrap it in torch.no_grad())", stacklevel=2)
        return self._c_forward(x, training=self.training)

    def _encoder_forward(self, x):
        if self._needs_graph():
            _, params = self._params_list("encoder")
            feats = list(_EncoderFn.apply(self, x, *params))
        else:
            feats = self._c_encoder_forward(x, training=self.training)
        return [x] + feats

    def _decoder_forward(self, *features):
        feats = list(features[1:])
        if self._needs_graph():
            _, params = self._params_list("decoder")
            return _DecoderFn.apply(self, *feats, *params)
        return self._c_decoder_forward(feats, training=self.training)

    def _head_forward(self, x):
        if self._needs_graph():
            _, params = self._params_list("segmentation_head")
            return _HeadFn.apply(self, x, *params)
        return self._c_head_forward(x, training=self.training)

    @torch.no_grad()
    def predict(self, x):
        was = self.training
        self.eval()
        try:
            return self.forward(x)
        finally:
            self.train(was)

    @torch.no_grad()
    @_on_model_device
    def predict_classes(self, x, want_prob=False):
        """Eval-mode forward whose head convolution returns the argmax class per pixel (uint8 (B,H,W)) and, with ``want_prob``,
        that class's softmax probability (fp32 (B,H,W)) from its epilogue — what ``convert(softmax(model(x)), 'argmax')`` of
        zone_detect (dataset.py:23-30) and ``predict_step`` (task_module.py:206-213) compute — without writing the logits."""
        x = self._prep(x)
        B, _, H, W = x.shape
        preds = torch.empty(B, H, W, dtype=torch.uint8, device=x.device)
        prob = torch.empty(B, H, W, dtype=torch.float32, device=x.device) if want_prob else None
        self._c_forward(x, training=False, want_logits=False, preds=preds, prob=prob)
        return (preds, prob) if want_prob else preds

    def __del__(self):
        try:
            for k in ("_h", "_h_eval"):
                h = self.__dict__.get(k)
                if h:
                    L.lib().flair_unet_destroy(h)
        except Exception:
            pass

    _NATIVE_STATE = ("_h", "_h_eval", "_eval_key", "_eval_graphs", "_eval_hits", "_ws_need", "_split", "_ws", "_flat_p", "_flat_b", "_flat_n", "_grads", "_live_ws", "_split_ws")

    def __getstate__(self):
        d = dict(self.__dict__)
        for k in self._NATIVE_STATE:
            d.pop(k, None)
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        h = C.c_void_p()
        L.check(L.lib().flair_unet_create(C.byref(h), self.in_channels, self.classes, self._dt), "flair_unet_create")
        object.__setattr__(self, "_h", h)
        self._flat_p = self._flat_b = self._flat_n = self._grads = None
        self._ws = {}
        self._live_id = -1
        for m in (self.encoder, self.decoder, self.segmentation_head):
            object.__setattr__(m, "_owner", self)


def create_model(arch, encoder_name="resnet34", encoder_weights="imagenet", in_channels=3, classes=1, **kwargs):
    """``smp.create_model`` for the one architecture the reference instantiates (src/flair/model.py:37-41)."""
    archs = {"unet": Unet}
    try:
        cls = archs[arch.lower()]
    except KeyError:
        raise KeyError(f"Wrong architecture type `{arch}`. Available options are: {list(archs.keys())}")
    return cls(encoder_name=encoder_name, encoder_weights=encoder_weights, in_channels=in_channels, classes=classes, **kwargs)
