#!/usr/bin/env python3
"""Element-level gradient golden for the reference's training step (the |.|-sum checksums of
step_c13_b2_64.npz cannot tell WHERE a gradient differs).

Replays tests/golden/make_golden.py's ``golden_step`` — the REFERENCE's own FLAIR_ModelFactory (src/flair/model.py) and
segmentation_task_training.step (src/flair/task_module.py), loaded by path with the same arithmetic-free stand-ins — checks
that loss and gradient checksums reproduce the committed step_c13_b2_64.npz exactly, and stores, per parameter tensor, the
gradient's max |.| and 48 elements at seeded positions (data only):

    python tests/golden/make_golden_grads.py        # build container only (needs /root/reference)
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402

NS = 48


def main():
    torch.set_num_threads(8)
    G._install_stubs()
    ref_model = G._load("ref_flair_model", "src/flair/model.py")
    ref_task = G._load("ref_flair_task_module", "src/flair/task_module.py")
    old = np.load(os.path.join(HERE, "step_c13_b2_64.npz"))
    C, CIN = 13, 5
    classes = {k: G.CLASSES19[k] for k in range(1, C + 1)}
    torch.manual_seed(int(old["seed"]))
    model = ref_model.FLAIR_ModelFactory(G._config(CIN, classes))
    img = torch.from_numpy(old["img"])
    lab = torch.from_numpy(old["lab"]).long()
    msk = torch.stack([(lab == i) for i in range(C)], dim=1).float()
    criterion = nn.CrossEntropyLoss(weight=torch.from_numpy(old["weight"]))
    task = ref_task.segmentation_task_training(model=model, class_infos=classes, criterion=criterion, optimizer=None,
                                               use_metadata=False, scheduler=None)
    task.train()
    loss, _, _ = task.step({"img": img, "msk": msk})
    loss.backward()
    assert abs(loss.item() - float(old["loss"])) == 0.0, (loss.item(), float(old["loss"]))
    named = dict(model.seg_model.named_parameters())
    for k, v in zip(old["grad_keys"], old["grad_abs_sum"]):
        assert float(named[str(k)].grad.double().abs().sum()) == float(v), k   # the same run as the committed checksums
    rng = np.random.default_rng(2022)
    keys, amax, idx, val = [], [], [], []
    for k, p in model.seg_model.named_parameters():
        g = p.grad.detach().flatten()
        n = g.numel()
        ii = np.sort(rng.choice(n, size=min(NS, n), replace=False))
        ii = np.pad(ii, (0, NS - len(ii)), constant_values=-1)
        keys.append(k)
        amax.append(float(g.abs().max()))
        idx.append(ii)
        val.append(np.where(ii >= 0, g.numpy()[np.maximum(ii, 0)], 0.0).astype(np.float32))
    np.savez_compressed(os.path.join(HERE, "step_c13_b2_64_grads.npz"), keys=np.array(keys), abs_max=np.array(amax, np.float64),
                        index=np.stack(idx).astype(np.int64), value=np.stack(val))
    print("tensors", len(keys), "samples", NS)


if __name__ == "__main__":
    main()
