#!/usr/bin/env python3
"""mIoU-parity sanity run (BASELINE.json: "... ; mIoU parity"): the same synthetic, learnable segmentation task trained
with the HIP path in fp32 (parity mode) and bf16 (throughput mode), plus the CPU oracle for a shorter prefix.
Tiles: 5 uint8 bands, label = quantised mix of two bands (so the net can learn it); device feed with 'custom'
normalisation and D4 augmentation; SGD lr 0.02 (the reference's default), batch 16.  Prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flair-1_amd"))
import numpy as np
import torch
import flair_amd
from flair_amd.data_feed import TileFeed

dev = torch.device("cuda:0")
S, C, NTRAIN, NVAL, BATCH = 256, 13, 96, 32, 16
STEPS = int(os.environ.get("STEPS", "240"))
MEANS, STDS = [105.08, 110.87, 101.82, 106.38, 53.26], [52.17, 45.38, 44, 39.69, 79.3]


def make_tiles(n, seed):
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(n, 5, S // 8, S // 8, generator=g)
    img = torch.nn.functional.interpolate(base, size=(S, S), mode="bilinear", align_corners=False)
    img = (img + 0.08 * torch.randn(n, 5, S, S, generator=g)).clamp(0, 1)
    lab = ((0.6 * img[:, 0] + 0.4 * img[:, 3]) * C).floor().clamp(0, C - 1).to(torch.uint8) + 1   # stored labels 1..C
    return (img * 255).to(torch.uint8), lab


def miou(cm):
    cm = cm.double()
    d = cm.diag()
    den = cm.sum(0) + cm.sum(1) - d
    return float((d[den > 0] / den[den > 0]).mean())


def hip_model(dtype):
    """the HIP model with the oracle's seeded weights (the two packages draw their initial weights in different orders)"""
    from oracle import unet_resnet34 as om
    m = flair_amd.create_model("unet", "resnet34", encoder_weights=None, in_channels=5, classes=C, compute_dtype=dtype)
    m.load_state_dict(om.seeded_model(5, C, 2022).state_dict(), strict=True)
    return m.to(dev).train()


def run_prefix(dtype, train, steps):
    """first losses without augmentation, same sampling order as run_cpu"""
    m = hip_model(dtype)
    tr = flair_amd.SegTrainer(m, lr=0.02)
    plain = TileFeed([1, 2, 3, 4, 5], C, "custom", MEANS, STDS)
    order = torch.Generator().manual_seed(11)
    out = []
    for s in range(steps):
        idx = torch.randperm(NTRAIN, generator=order)[:BATCH]
        b = plain(train[0][idx].to(dev), train[1][idx].to(dev))
        out.append(round(tr.train_step(b["img"], b["msk"]).item(), 4))
    return out


def run(dtype, train, val, steps):
    m = hip_model(dtype)
    tr = flair_amd.SegTrainer(m, lr=0.02)
    feed = TileFeed([1, 2, 3, 4, 5], C, "custom", MEANS, STDS, use_augmentations=True, generator=torch.Generator().manual_seed(7))
    plain = TileFeed([1, 2, 3, 4, 5], C, "custom", MEANS, STDS)
    order = torch.Generator().manual_seed(11)
    losses = []
    t0 = time.time()
    for s in range(steps):
        idx = torch.randperm(NTRAIN, generator=order)[:BATCH]
        b = feed(train[0][idx].to(dev), train[1][idx].to(dev))
        losses.append(tr.train_step(b["img"], b["msk"]).item())
    cm = torch.zeros(C, C, dtype=torch.int64, device=dev)
    from flair_amd import ops
    for i in range(0, NVAL, BATCH):
        b = plain(val[0][i:i + BATCH].to(dev), val[1][i:i + BATCH].to(dev))
        ops.confmat_update(cm, b["msk"], tr.predict(b["img"]))
    return {"loss_first": round(float(np.mean(losses[:5])), 4), "loss_last": round(float(np.mean(losses[-10:])), 4),
            "val_miou": round(miou(cm.cpu()), 4), "seconds": round(time.time() - t0, 1)}


def run_cpu(train, steps):
    from oracle import seg_step, unet_resnet34 as om
    from oracle import data_feed as F
    m = om.seeded_model(5, C, 2022).train()
    order = torch.Generator().manual_seed(11)
    losses = []
    for s in range(steps):
        idx = torch.randperm(NTRAIN, generator=order)[:BATCH]
        x = torch.stack([torch.from_numpy(F.norm_np(train[0][i].numpy(), "custom", MEANS, STDS)) for i in idx])
        lab = torch.stack([torch.from_numpy(F.labels_from_raw(train[1][i].numpy(), C)) for i in idx]).long()
        m.zero_grad()
        loss = torch.nn.functional.cross_entropy(m(x), lab)
        loss.backward()
        seg_step.sgd_step_(m.parameters(), 0.02)
        losses.append(loss.item())
    return [round(v, 4) for v in losses]


if __name__ == "__main__":
    train, val = make_tiles(NTRAIN, 1), make_tiles(NVAL, 2)
    out = {"task": f"{NTRAIN} train / {NVAL} val tiles 5x{S}x{S}, {C} classes, batch {BATCH}, SGD lr 0.02, {STEPS} steps",
           "hip_f32": run("f32", train, val, STEPS), "hip_bf16": run("bf16", train, val, STEPS)}
    if os.environ.get("CPU_STEPS", "0") != "0":
        n = int(os.environ["CPU_STEPS"])
        torch.set_num_threads(16)
        # same order, no augmentation on either side for the short prefix comparison
        out["cpu_oracle_losses_no_aug"] = run_cpu(train, n)
        out["hip_f32_losses_no_aug"] = run_prefix("f32", train, n)
        out["hip_bf16_losses_no_aug"] = run_prefix("bf16", train, n)
    print(json.dumps(out))
