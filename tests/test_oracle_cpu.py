"""CPU suite (-m "not gpu"): the oracle against the golden vectors produced from the reference's own
files (tests/golden/make_golden.py), the known-answer facts of SURVEY.md §4, and the host logic."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import seg_step
from oracle import unet_resnet34 as om


def test_param_counts_known_answers():
    # README.md:91 "about 24.4M"; smp's documented 24,436,369 for Unet(resnet34, 3ch, 1 class)
    for cin, cls, n in ((3, 1, 24_436_369), (5, 13, 24_444_381), (5, 19, 24_445_251)):
        m = om.create_model("unet", "resnet34", in_channels=cin, classes=cls)
        assert sum(p.numel() for p in m.parameters()) == n
    enc = sum(p.numel() for p in m.encoder.parameters())
    dec = sum(p.numel() for p in m.decoder.parameters())
    assert dec == 3_151_552 and enc == 21_284_672 + (5 - 3) * 64 * 49


def test_state_dict_keys_golden(golden_dir):
    keys = json.load(open(os.path.join(golden_dir, "state_dict_keys_c3_cls1.json")))
    sd = om.create_model("unet", "resnet34", in_channels=3, classes=1).state_dict()
    assert [k for k, _ in keys] == list(sd.keys())
    assert all(list(sd[k].shape) == s for k, s in keys)
    assert "encoder.layer2.0.downsample.0.weight" in sd and "decoder.blocks.4.conv2.1.running_var" in sd
    assert "segmentation_head.0.bias" in sd and not any(k.startswith("encoder.fc") for k in sd)


def test_create_model_errors_and_shape_check():
    with pytest.raises(KeyError):
        om.create_model("fpn")
    with pytest.raises(KeyError):
        om.create_model("unet", "resnet50")
    m = om.create_model("unet", "resnet34", in_channels=5, classes=13)
    with pytest.raises(RuntimeError, match="divisible by 32"):
        m(torch.zeros(1, 5, 70, 64))
    feats = m.encoder(torch.zeros(1, 5, 64, 64))
    assert [f.shape[1] for f in feats] == [5, 64, 64, 128, 256, 512] and feats[-1].shape[-1] == 2


def test_metrics_against_reference_outputs(golden_dir):
    """class_IoU / OA / precision / recall / F-score: values computed by /root/reference/src/flair/metrics.py."""
    gold = json.load(open(os.path.join(golden_dir, "metrics_reference.json")))
    g = gold["c13_seed0"]
    rng = np.random.default_rng(0)
    t = rng.integers(0, 13, size=512 * 512)
    p = np.where(rng.random(t.size) < 0.7, t, rng.integers(0, 13, size=t.size))
    cm = seg_step.confusion_matrix_np(t, p, 13)
    assert np.array_equal(cm, np.array(g["confmat"]))
    iou, miou = seg_step.class_IoU(cm, 13)
    assert miou == g["miou"] == 56.541990939912836  # SURVEY.md §4 known answer
    assert seg_step.overall_accuracy(cm) == g["oa"] == 72.23930358886719
    pr, mp = seg_step.class_precision(cm)
    rc, mr = seg_step.class_recall(cm)
    fs, mf = seg_step.class_fscore(pr, rc)
    assert mf == g["mfscore"] == 72.23816436846431
    assert np.array_equal(iou, g["iou"]) and np.array_equal(pr, g["precision"]) and np.array_equal(rc, g["recall"])
    g = gold["c19_seed1_zero_weight"]
    m = seg_step.offline_metrics(np.array(g["confmat"]), g["weights"])
    assert m["Avg_metrics"] == [g["miou"], g["oa"], g["mfscore"], g["mprecision"], g["mrecall"]]
    assert m["per_class_iou"] == g["iou"] and 0.0 in g["iou"]  # absent class: 0/0 -> NaN -> 0 (metrics.py:21)
    assert len(g["iou"]) == 15  # 4 weight-0 classes removed (metrics.py:77-82)


def test_jaccard_restatement(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "jaccard_restated.json")))
    cm = seg_step.confusion_matrix_np(np.array(g["target"]), np.array(g["pred"]), 13)
    assert np.array_equal(cm, np.array(g["confmat"]))
    per = seg_step.jaccard_from_confmat(cm, None)
    assert np.allclose(per, g["none"]) and per[7] == 0.0  # class absent from target and pred
    assert abs(seg_step.jaccard_from_confmat(cm, "weighted") - g["weighted"]) < 1e-12
    # hand check of the definitions on a 2x2 case
    cm2 = np.array([[3, 1], [2, 4]])
    assert np.allclose(seg_step.jaccard_from_confmat(cm2, None), [3 / 6, 4 / 7])
    assert abs(seg_step.jaccard_from_confmat(cm2, "weighted") - (4 * 0.5 + 6 * 4 / 7) / 10) < 1e-12


def test_step_golden_reproduced_by_oracle(golden_dir):
    """The oracle alone (no reference import) reproduces the vector made by the reference's step()."""
    g = np.load(os.path.join(golden_dir, "step_c13_b2_64.npz"))
    torch.manual_seed(int(g["seed"]))
    model = om.create_model("unet", "resnet34", in_channels=5, classes=13)
    img = torch.from_numpy(g["img"])
    lab = torch.from_numpy(g["lab"]).long()
    msk = torch.stack([(lab == i) for i in range(13)], 1).float()
    w = torch.from_numpy(g["weight"])
    model.train()
    logits = model(img)
    loss, preds, targets = seg_step.step_torch(logits, msk, w)
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    assert torch.equal(preds.to(torch.uint8), torch.from_numpy(g["preds"]))
    assert torch.equal(targets.to(torch.uint8), torch.from_numpy(g["targets"]))
    assert (targets.view(2, 64, 64)[0, :4, :4] == 0).all()  # label 200 -> all-zero one-hot -> class 0
    assert np.abs(logits.detach().numpy() - g["logits"]).max() < 1e-5
    loss_np, dl = seg_step.cross_entropy_np(logits.detach().numpy(), targets.view(2, 64, 64).long().numpy(), w.numpy())
    assert abs(loss_np - float(g["loss"])) < 1e-6
    loss.backward()
    named = dict(model.named_parameters())
    for k, v in zip(g["grad_keys"], g["grad_abs_sum"]):
        assert abs(float(named[str(k)].grad.double().abs().sum()) - v) <= 1e-5 * abs(v) + 1e-9


def test_predict_convert_and_sgd():
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(1, 19, 8, 8, generator=g)
    p = seg_step.predict_torch(logits)
    assert p.dtype == torch.int64 and torch.equal(p, logits.argmax(1))
    conv = seg_step.detect_convert_argmax(torch.softmax(logits, 1)[0].numpy())
    assert conv.shape == (2, 8, 8) and conv.dtype == np.float32
    lin = nn.Linear(3, 2)
    lin(torch.ones(1, 3)).sum().backward()
    before = [q.detach().clone() for q in lin.parameters()]
    seg_step.sgd_step_(lin.parameters(), 0.5)
    for b, q in zip(before, lin.parameters()):
        assert torch.allclose(q, b - 0.5 * q.grad)


def test_weighted_ce_zero_weight_classes():
    g = torch.Generator().manual_seed(1)
    logits = torch.randn(2, 19, 6, 6, generator=g)
    lab = torch.randint(0, 19, (2, 6, 6), generator=g)
    w = torch.tensor([1.0] * 14 + [0, 0, 0, 1, 0])  # configs/flair-1-config.yaml:57-75
    ref = nn.functional.cross_entropy(logits, lab, weight=w)
    loss, dl = seg_step.cross_entropy_np(logits.numpy(), lab.numpy(), w.numpy())
    assert abs(loss - ref.item()) < 1e-6
    lg = logits.clone().requires_grad_(True)
    nn.functional.cross_entropy(lg, lab, weight=w).backward()
    assert np.abs(dl - lg.grad.numpy()).max() < 1e-7


def test_encoder_restatement_matches_transformers_resnet34(golden_dir):
    """Two-source check of the ResNet34 trunk (smp / torchvision are absent; `transformers` has an independent
    implementation of the same network): the oracle's encoder features equal those of transformers.ResNetModel carrying
    the oracle's weights (tests/golden/make_golden_hf_resnet.py), in eval mode and with batch statistics."""
    import numpy as np
    import torch
    from oracle import unet_resnet34 as om
    g = np.load(os.path.join(golden_dir, "hf_resnet34_features.npz"))
    ref = om.seeded_model(5, 13, seed=int(g["seed_model"]))
    x = torch.from_numpy(g["x"])
    for mode in ("eval", "train"):
        ref.train(mode == "train")
        with torch.no_grad():
            feats = ref.encoder(x)
        assert len(feats) == 6 and feats[0] is x or torch.equal(feats[0], x)
        for i in range(1, 6):
            want = torch.from_numpy(g[f"{mode}_f{i}"])
            assert feats[i].shape == want.shape
            err = float((feats[i] - want).abs().max())
            assert err <= 2e-5 * max(1.0, float(want.abs().max())), (mode, i, err)
    # and against the live model where transformers is importable (the build container and the GPU image both have it)
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location("mk_hf", os.path.join(golden_dir, "make_golden_hf_resnet.py"))
        mk = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mk)
        hf = mk.hf_resnet34(5)
    except Exception:  # noqa: BLE001
        return
    hf.load_state_dict(mk.remap(ref.state_dict()), strict=False)
    hf.eval(); ref.eval()
    x2 = torch.randn(1, 5, 96, 64, generator=torch.Generator().manual_seed(11))
    with torch.no_grad():
        a, b = ref.encoder(x2)[1:], mk.hf_features(hf, x2)
    assert all(float((u - v).abs().max()) <= 2e-5 * max(1.0, float(v.abs().max())) for u, v in zip(a, b))
