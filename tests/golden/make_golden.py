#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own files.

Run in the build container only (needs /root/reference, read-only):

    python tests/golden/make_golden.py

What is executed from the reference (unchanged, loaded by path):
  * src/flair/metrics.py      overall_accuracy / class_IoU / class_precision / class_recall / class_fscore
  * src/flair/model.py        FLAIR_ModelFactory.__init__/forward, MetadataMLP
  * src/flair/task_module.py  segmentation_task_training.step, segmentation_task_predict.predict_step

Third-party packages the reference imports but that are absent here are replaced in
``sys.modules`` by minimal stand-ins that carry NO arithmetic of the path:
pytorch_lightning (LightningModule = nn.Module, rank_zero_only = identity),
torchmetrics (empty classes; never called by step()), transformers (names only).
``segmentation_models_pytorch.create_model`` is served by oracle/unet_resnet34.py — the
restated third-party model (PARITY UNPINNED for that part, see oracle/__init__.py).

Outputs are DATA only (inputs + expected outputs); no reference source text is stored.
"""
import hashlib
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import unet_resnet34 as oracle_model  # noqa: E402
from oracle import seg_step  # noqa: E402


def _install_stubs():
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = nn.Module
    plu = types.ModuleType("pytorch_lightning.utilities")
    plr = types.ModuleType("pytorch_lightning.utilities.rank_zero")
    plr.rank_zero_only = lambda f: f
    sys.modules.update({"pytorch_lightning": pl, "pytorch_lightning.utilities": plu,
                        "pytorch_lightning.utilities.rank_zero": plr})
    tm = types.ModuleType("torchmetrics")
    tmc = types.ModuleType("torchmetrics.classification")
    tma = types.ModuleType("torchmetrics.aggregation")
    tmc.MulticlassJaccardIndex = type("MulticlassJaccardIndex", (), {})
    tma.MeanMetric = type("MeanMetric", (), {})
    sys.modules.update({"torchmetrics": tm, "torchmetrics.classification": tmc, "torchmetrics.aggregation": tma})
    tr = types.ModuleType("transformers")
    tr.AutoModelForSemanticSegmentation = object
    tr.AutoConfig = object
    sys.modules["transformers"] = tr
    smp = types.ModuleType("segmentation_models_pytorch")
    smp.create_model = oracle_model.create_model
    sys.modules["segmentation_models_pytorch"] = smp


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


CLASSES19 = {1: [1, 'building'], 2: [1, 'pervious surface'], 3: [1, 'impervious surface'], 4: [1, 'bare soil'],
             5: [1, 'water'], 6: [1, 'coniferous'], 7: [1, 'deciduous'], 8: [1, 'brushwood'], 9: [1, 'vineyard'],
             10: [1, 'herbaceous vegetation'], 11: [1, 'agricultural land'], 12: [1, 'plowed land'],
             13: [1, 'swimming_pool'], 14: [1, 'snow'], 15: [0, 'clear cut'], 16: [0, 'mixed'], 17: [0, 'ligneous'],
             18: [1, 'greenhouse'], 19: [0, 'other']}  # configs/flair-1-config.yaml:56-75


def golden_metrics(ref_metrics):
    out = {}
    # SURVEY.md §4 recipe (13 classes)
    rng = np.random.default_rng(0)
    t = rng.integers(0, 13, size=512 * 512)
    p = np.where(rng.random(t.size) < 0.7, t, rng.integers(0, 13, size=t.size))
    from sklearn.metrics import confusion_matrix
    cm = confusion_matrix(t, p, labels=list(range(13)))
    iou, miou = ref_metrics.class_IoU(cm, 13)
    oa = ref_metrics.overall_accuracy(cm)
    pr, mpr = ref_metrics.class_precision(cm)
    rc, mrc = ref_metrics.class_recall(cm)
    fs, mfs = ref_metrics.class_fscore(pr, rc)
    out["c13_seed0"] = {"recipe": "rng=default_rng(0); t=integers(0,13,512*512); p=where(random<0.7,t,integers(0,13))",
                        "confmat": cm.tolist(), "iou": iou.tolist(), "miou": float(miou), "oa": float(oa),
                        "precision": pr.tolist(), "mprecision": float(mpr), "recall": rc.tolist(),
                        "mrecall": float(mrc), "fscore": fs.tolist(), "mfscore": float(mfs)}
    # 19 classes, weight-0 removal (metrics.py:77-88), absent classes (0/0 -> NaN -> 0)
    rng = np.random.default_rng(1)
    t = rng.integers(0, 19, size=256 * 256)
    t[t == 13] = 0  # class 14 'snow' never occurs in target
    p = np.where(rng.random(t.size) < 0.6, t, rng.integers(0, 19, size=t.size))
    p[p == 13] = 1  # ... nor in pred -> 0/0
    p[p == 8] = 2   # class 9 never predicted -> precision 0/0
    cm = confusion_matrix(t, p, labels=list(range(19)))
    weights = np.array([CLASSES19[i][0] for i in CLASSES19])
    unused = np.where(weights == 0)[0]
    cmc = np.delete(np.delete(cm, unused, axis=0), unused, axis=1)
    with np.errstate(all="ignore"):
        iou, miou = ref_metrics.class_IoU(cmc, len(np.nonzero(weights)[0]))
        oa = ref_metrics.overall_accuracy(cmc)
        pr, mpr = ref_metrics.class_precision(cmc)
        rc, mrc = ref_metrics.class_recall(cmc)
        fs, mfs = ref_metrics.class_fscore(pr, rc)
    out["c19_seed1_zero_weight"] = {"confmat": cm.tolist(), "weights": weights.tolist(), "iou": iou.tolist(),
                                    "miou": float(miou), "oa": float(oa), "precision": pr.tolist(),
                                    "mprecision": float(mpr), "recall": rc.tolist(), "mrecall": float(mrc),
                                    "fscore": fs.tolist(), "mfscore": float(mfs)}
    with open(os.path.join(HERE, "metrics_reference.json"), "w") as f:
        json.dump(out, f)
    print("metrics:", out["c13_seed0"]["miou"], out["c13_seed0"]["oa"], out["c13_seed0"]["mfscore"])


def _config(n_ch, classes, use_metadata=False):
    return {"model_framework": {"model_provider": "SegmentationModelsPytorch",
                                "SegmentationModelsPytorch": {"encoder_decoder": "resnet34_unet"}},
            "use_metadata": use_metadata, "channels": list(range(1, n_ch + 1)), "classes": classes}


def _param_checksums(model):
    return {k: float(v.double().abs().sum()) for k, v in model.state_dict().items() if v.dtype.is_floating_point}


def golden_step(ref_model, ref_task):
    """One train step through the reference's FLAIR_ModelFactory + step(); then predict_step."""
    C, CIN, B, H = 13, 5, 2, 64
    classes = {k: CLASSES19[k] for k in range(1, C + 1)}
    torch.manual_seed(2022)
    model = ref_model.FLAIR_ModelFactory(_config(CIN, classes))  # -> oracle create_model under seed 2022
    n_params = sum(p.numel() for p in model.seg_model.parameters())
    assert n_params == 24_444_381, n_params
    g = torch.Generator().manual_seed(7)
    img = torch.randn(B, CIN, H, H, generator=g)
    lab = torch.randint(0, C, (B, H, H), generator=g)
    lab[0, :4, :4] = 200  # labels >= C give an all-zero one-hot -> class 0 (data_loader.py:67-68, task_module.py:71)
    msk = torch.stack([(lab == i) for i in range(C)], dim=1).float()
    weight = torch.linspace(0.5, 1.5, C)
    weight[3] = 0.0
    criterion = nn.CrossEntropyLoss(weight=weight)
    task = ref_task.segmentation_task_training(model=model, class_infos=classes, criterion=criterion,
                                               optimizer=None, use_metadata=False, scheduler=None)
    task.train()
    loss, preds, targets = task.step({"img": img, "msk": msk})
    logits = task.forward(img, "").detach()  # second fwd: same batch stats, running stats updated twice
    loss.backward()
    grads = {k: float(p.grad.double().abs().sum()) for k, p in model.seg_model.named_parameters()}
    # plain SGD lr 0.02 (tasks_utils.py:95, flair-1-config.yaml:50)
    opt = torch.optim.SGD(model.parameters(), lr=0.02)
    opt.step()
    post = _param_checksums(model.seg_model)
    np.savez_compressed(
        os.path.join(HERE, "step_c13_b2_64.npz"),
        img=img.numpy(), lab=lab.numpy().astype(np.uint8), weight=weight.numpy(),
        loss=np.float64(loss.item()), preds=preds.numpy().astype(np.uint8), targets=targets.numpy().astype(np.uint8),
        logits=logits.numpy(),
        grad_keys=np.array(list(grads.keys())), grad_abs_sum=np.array(list(grads.values())),
        post_keys=np.array(list(post.keys())), post_abs_sum=np.array(list(post.values())),
        seed=np.int64(2022), lr=np.float64(0.02), n_params=np.int64(n_params),
    )
    print("step: loss", loss.item(), "preds sha", _sha(preds.numpy().astype(np.uint8))[:12])

    # predict_step (eval mode, running stats after the 2 forwards above) on one 512x512 tile
    ptask = ref_task.segmentation_task_predict(model=model, num_classes=C, use_metadata=False)
    ptask.eval()
    g = torch.Generator().manual_seed(11)
    tile = torch.randn(1, CIN, 512, 512, generator=g)
    with torch.no_grad():
        out = ptask.predict_step({"img": tile, "id": ["x"]}, 0)
        lg = ptask.forward(tile, "")
    pm = out["preds"].numpy().astype(np.uint8)
    np.savez_compressed(
        os.path.join(HERE, "predict_c13_512.npz"),
        tile_seed=np.int64(11), preds=pm, logits_crop=lg[0, :, 240:272, 240:272].numpy(),
        logits_mean=lg.double().mean(dim=(0, 2, 3)).numpy(), logits_std=lg.double().std(dim=(0, 2, 3)).numpy(),
        preds_sha256=np.array(_sha(pm)),
    )
    print("predict: sha", _sha(pm)[:12], "keys", sorted(out.keys()))


def golden_metadata(ref_model):
    """Metadata fusion path (model.py:57-62).  The reference raises NameError at model.py:32
    (bare ``model_provider``); the evident intent is restored by defining that module global."""
    C, CIN = 19, 5
    ref_model.model_provider = "SegmentationModelsPytorch"
    torch.manual_seed(2023)
    model = ref_model.FLAIR_ModelFactory(_config(CIN, CLASSES19, use_metadata=True))
    del ref_model.model_provider
    n_params = sum(p.numel() for p in model.seg_model.parameters())
    assert n_params == 24_445_251, n_params
    model.eval()  # dropout off (Q5), BN running stats (init 0/1)
    g = torch.Generator().manual_seed(5)
    tile = torch.randn(1, CIN, 512, 512, generator=g)
    mtd = torch.rand(1, 45, generator=g)
    with torch.no_grad():
        lg = model(tile, mtd)
        enc = model.enc(mtd)
    np.savez_compressed(
        os.path.join(HERE, "metadata_c19_512.npz"),
        tile_seed=np.int64(5), mtd=mtd.numpy(), x_enc=enc.numpy(),
        logits_crop=lg[0, :, 100:132, 300:332].numpy(),
        logits_mean=lg.double().mean(dim=(0, 2, 3)).numpy(),
        preds_sha256=np.array(_sha(lg.argmax(1).numpy().astype(np.uint8))),
    )
    print("metadata: logits mean", lg.mean().item())


def golden_keys():
    m = oracle_model.seeded_model(3, 1, seed=0)
    assert sum(p.numel() for p in m.parameters()) == 24_436_369
    keys = [[k, list(v.shape)] for k, v in m.state_dict().items()]
    with open(os.path.join(HERE, "state_dict_keys_c3_cls1.json"), "w") as f:
        json.dump(keys, f)
    print("keys:", len(keys))


if __name__ == "__main__":
    torch.set_num_threads(8)
    _install_stubs()
    ref_metrics = _load("ref_flair_metrics", "src/flair/metrics.py")
    ref_model = _load("ref_flair_model", "src/flair/model.py")
    ref_task = _load("ref_flair_task_module", "src/flair/task_module.py")
    golden_metrics(ref_metrics)
    golden_keys()
    golden_step(ref_model, ref_task)
    golden_metadata(ref_model)
    # torchmetrics is absent: Jaccard goldens come from the oracle restatement (published algorithm) — flagged
    rng = np.random.default_rng(3)
    t = rng.integers(0, 13, size=4096)
    t[t == 7] = 2
    p = np.where(rng.random(t.size) < 0.5, t, rng.integers(0, 13, size=t.size))
    p[p == 7] = 3
    cm = seg_step.confusion_matrix_np(t, p, 13)
    with open(os.path.join(HERE, "jaccard_restated.json"), "w") as f:
        json.dump({"note": "torchmetrics-1.2.0 absent: values from oracle/seg_step.py restatement (PARITY UNPINNED)",
                   "target": t.tolist(), "pred": p.tolist(), "confmat": cm.tolist(),
                   "weighted": seg_step.jaccard_from_confmat(cm, "weighted"),
                   "none": seg_step.jaccard_from_confmat(cm, None).tolist(),
                   "macro": float(seg_step.jaccard_from_confmat(cm, "macro"))}, f)
