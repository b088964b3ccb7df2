"""Fused training step + data-parallel gradient exchange for the segmentation hot path.

One step = what Lightning drives per batch in the reference:
``training_step`` (src/flair/task_module.py:82-86: forward, CE, argmax, Jaccard update) ->
``loss.backward()`` -> DDP gradient all-reduce (src/flair/tasks.py:83-88, strategy 'ddp') ->
``SGD.step()`` (src/flair/tasks_utils.py:95).  Here every device-side piece is a HIP kernel behind the
C ABI and the exchange is RCCL over xGMI: one process per GPU, gradients live in ONE flat buffer laid
out stage by stage, the native backward records a HIP event when a stage's bucket is final, and the
bucket's all-reduce starts on a side stream while the earlier stages are still being differentiated.
BatchNorm batch statistics stay per rank (no SyncBN), like the reference; the RUNNING statistics are rank 0's on every
rank, like torch DDP's ``broadcast_buffers=True`` (the default Lightning's 'ddp' strategy keeps) leaves them: DDP sends rank
0's buffers to every rank in front of each forward, which for BatchNorm is the same as sending them behind each training
step — done here with one small broadcast per step on the exchange stream — so validation, predict, ``state_dict()`` and
checkpoints see rank 0's statistics on every rank without a collective of their own.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import _lib as L
from . import ops


def bucket_ranges(stage_ranges, min_elems=0):
    """Gradient buckets in READY order (head first).  Small neighbours are merged so that no bucket is
    below ``min_elems`` floats: xGMI rings are per-link bound, tiny collectives only add latency."""
    out = []
    cur = None
    for b, e in reversed(list(stage_ranges)):
        if e <= b:
            continue
        cur = (b, e) if cur is None else (b, cur[1])
        if cur[1] - cur[0] >= min_elems:
            out.append(cur)
            cur = None
    if cur is not None:
        if out:
            out[-1] = (cur[0], out[-1][1])
        else:
            out.append(cur)
    return out


def allreduce_buckets(flat_grads, buckets, group=None, async_op=False):
    """Sum-all-reduce each bucket view of the flat gradient buffer (works on gloo/CPU for tests)."""
    works = []
    for b, e in buckets:
        w = dist.all_reduce(flat_grads[b:e], op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            works.append(w)
    return works


def shard_indices(n_items, rank, world, epoch_seed=0, shuffle=True, drop_last=False):
    """The partition Lightning's DistributedSampler gives the reference: a seeded
    permutation, padded to a multiple of ``world`` by wrapping, rank r takes r, r+W, r+2W, ..."""
    g = torch.Generator().manual_seed(epoch_seed)
    idx = torch.randperm(n_items, generator=g).tolist() if shuffle else list(range(n_items))
    if drop_last:
        idx = idx[: (n_items // world) * world]
    else:
        total = -(-n_items // world) * world
        idx = idx + idx[: total - len(idx)]
    return idx[rank::world]


class SegTrainer:
    """Owns the flat gradient buffer, the fused head outputs and the RCCL exchange for one rank."""

    def __init__(self, model, lr, class_weight=None, group=None, overlap=True, min_bucket_elems=1 << 18,
                 force_exchange=False, broadcast_buffers=True):
        self.model = model
        self.sync_buffers = bool(broadcast_buffers)   # torch DDP's broadcast_buffers (tasks.py:83-88 keeps the default, True)
        self._src0 = 0 if group is None else dist.get_global_rank(group, 0)
        self.lr = float(lr)
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        # rehearsal switch: run the event / side-stream / RCCL exchange even with one rank (1-GPU test boxes)
        self.exchange = self.world > 1 or (force_exchange and dist.is_available() and dist.is_initialized())
        self._no_collective = self.world == 1 and os.environ.get("FLAIR_REHEARSE_NO_COLLECTIVE") == "1"

        self.overlap = overlap
        p = model.flat_parameters()
        dev = p.device
        self.grads = torch.zeros_like(p)
        self.class_weight = None if class_weight is None else torch.as_tensor(class_weight, dtype=torch.float32, device=dev)
        self.confmat = torch.zeros(model.classes, model.classes, dtype=torch.int64, device=dev)
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self.buckets = bucket_ranges(model.stage_ranges(), min_bucket_elems)
        self._dl = None
        self._preds = None
        self._ce_ws = None
        if self.exchange:
            # HIGH priority: its own hardware queues (HIP multiplexes streams over a few queues per priority level; a
            # normal-priority stream can land on the caller's queue and serialise the collectives behind the backward
            # kernels they are meant to overlap), and a bucket's all-reduce is dispatched as soon as its stage event fires
            self.comm_stream = torch.cuda.Stream(device=dev, priority=-1)
            self.events = [torch.cuda.Event() for _ in range(7)]
            for e in self.events:
                e.record()  # materialise the hipEvent_t handles
            # bucket -> the stage whose completion makes it ready (its lowest stage)
            sr = model.stage_ranges()
            self._bucket_stage = [min(s for s, (b, e) in enumerate(sr) if b >= bb and e <= be and e > b) for bb, be in self.buckets]
            dist.broadcast(p, src=self._src0, group=group)  # DDP's initial parameter broadcast
            dist.broadcast(model.flat_buffers(), src=self._src0, group=group)

    def _buffers(self, B, H, W):
        m = self.model
        ld = L.lib().flair_unet_head_ld(m._h)
        dt = L.torch_dtype(m._dt)
        if self._dl is None or self._dl.shape != (B * H * W, ld):
            dev = self.grads.device
            self._dl = torch.empty(B * H * W, ld, dtype=dt, device=dev)
            self._preds = torch.empty(B, H, W, dtype=torch.uint8, device=dev)
            self._ce_ws = torch.empty(L.lib().flair_ce_workspace_bytes(B, H, W) + 256, dtype=torch.uint8, device=dev)
        return ld

    def train_step(self, img, labels):
        """img (B,Cin,H,W) fp32 HIP tensor; labels (B,H,W) uint8/int or fp32 one-hot (B,C,H,W).  Returns the
        device scalar loss (no host sync)."""
        with torch.cuda.device(self.grads.device):   # every native call below launches on the current device's stream
            return self._train_step(img, labels)

    def _train_step(self, img, labels):
        m = self.model
        B, _, H, W = img.shape
        ld = self._buffers(B, H, W)
        m._c_forward(img, training=True, want_logits=False)   # logits stay in the workspace, NHWC of the compute dtype
        kind = 3 if labels.dtype == torch.float32 else ops._LABEL_KIND[labels.dtype]
        l = L.lib()
        L.check(l.flair_ce_head_nhwc(l.flair_unet_logits_nhwc(m._h), m._dt, ld, L.ptr(labels.contiguous()), kind,
                                     L.ptr(self.class_weight), B, m.classes, H, W, L.ptr(self.loss), L.ptr(self._dl),
                                     L.ptr(self._preds), None, L.ptr(self.confmat), L.ptr(self._ce_ws), L.stream()),
                "ce_head_nhwc")
        if self.exchange and self.overlap:
            m._c_backward(dlogits_nhwc=self._dl, grads=self.grads, stage_events=self.events)
            works = []
            with torch.cuda.stream(self.comm_stream):
                for i, ((b, e), st) in enumerate(zip(self.buckets, self._bucket_stage)):
                    self.comm_stream.wait_event(self.events[st])
                    if self._no_collective:   # diagnostic (FLAIR_REHEARSE_NO_COLLECTIVE=1): stage events and waits only
                        continue
                    if i == 0 and self.sync_buffers:
                        # the first stage event lies behind the whole forward: this step's running statistics are final
                        works.append(dist.broadcast(m.flat_buffers(), src=self._src0, group=self.group, async_op=True))
                    works.append(dist.all_reduce(self.grads[b:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            for w in works:
                w.wait()
        else:
            m._c_backward(dlogits_nhwc=self._dl, grads=self.grads)
            if self.exchange:
                allreduce_buckets(self.grads, self.buckets, self.group)
                if self.sync_buffers:
                    dist.broadcast(m.flat_buffers(), src=self._src0, group=self.group)
        # DDP averages: fold 1/world into the step size
        ops.sgd_step_(m._flat_p, self.grads, self.lr / self.world)
        m._native_writes = m.__dict__.get("_native_writes", 0) + 1   # parameters changed behind torch's version counter
        return self.loss

    @torch.no_grad()
    def predict(self, img):
        """predict_step of the reference (task_module.py:206-213): (B,H,W) uint8 argmax(softmax(logits))."""
        m = self.model
        B, _, H, W = img.shape
        with torch.cuda.device(self.grads.device):
            return self._predict(m, img, B, H, W)

    def _predict(self, m, img, B, H, W):
        preds = torch.empty(B, H, W, dtype=torch.uint8, device=img.device)
        # the argmax comes out of the head convolution's epilogue (or, for shapes its persistent kernel does not take, from the
        # separate pass over the NHWC logits inside the same native call): the logits never reach HBM
        m._c_forward(img, training=False, want_logits=False, preds=preds)
        return preds
