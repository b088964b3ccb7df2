"""Checkpoint interop — pure host logic, runs without a GPU: smp-0.3.3 key names round-trip
through .pth / Lightning-style .ckpt files, the class-count surgery of src/flair/main.py:111-138 and the prefix
stripping of src/zone_detect/model.py:72-75."""
import os

import pytest
import torch

from oracle import unet_resnet34 as om


def _classes(C, zero=()):
    return {i + 1: [0 if i in zero else 1, f"class{i + 1}"] for i in range(C)}


def _config(C, n_ch=5, **kw):
    cfg = {"model_framework": {"model_provider": "SegmentationModelsPytorch",
                               "SegmentationModelsPytorch": {"encoder_decoder": "resnet34_unet", "encoder_weights": None}},
           "use_metadata": False, "use_weights": True, "channels": list(range(1, n_ch + 1)), "classes": _classes(C, zero=(2,)),
           "learning_rate": 0.02}
    cfg.update(kw)
    return cfg


def test_oracle_state_dict_loads_from_pth_and_ckpt(tmp_path):
    from flair_amd import checkpoint as ck
    ref = om.seeded_model(5, 13, seed=3)
    sd = ref.state_dict()
    pth = str(tmp_path / "unet.pth")
    torch.save(sd, pth)
    model = ck.load_model(_config(13, model_weights=pth))
    got = model.state_dict()
    assert list(got) == list(sd)
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    # Lightning layout: every key under model.seg_model. (+ criterion.weight, dropped by the prefix filter)
    ckpt = str(tmp_path / "last.ckpt")
    lsd = {"model.seg_model." + k: v for k, v in sd.items()}
    lsd["criterion.weight"] = torch.ones(13)
    torch.save({"state_dict": lsd, "epoch": 4}, ckpt)
    weights = ck.get_module(ckpt)
    assert "criterion.weight" not in weights and list(weights) == list(sd)
    model2 = ck.load_model(_config(13, model_weights=ckpt))
    assert all(torch.equal(model2.state_dict()[k], sd[k]) for k in sd)
    assert ck.get_module(str(tmp_path / "absent.pth")) == {}
    with pytest.raises(RuntimeError):  # strict load of a 19-class file into a 13-class model
        torch.save(om.seeded_model(5, 19, seed=3).state_dict(), pth)
        ck.load_model(_config(13, model_weights=pth))


def test_load_checkpoint_class_count_surgery(tmp_path):
    from flair_amd import checkpoint as ck
    from flair_amd.tasks_utils import get_segmentation_module
    src = get_segmentation_module(_config(19), "train")
    with torch.no_grad():
        for p in src.parameters():
            p.add_(0.01)
    path = str(tmp_path / "c19.ckpt")
    ck.save_checkpoint(path, src, src.optimizer, src.scheduler, epoch=7, global_step=123)
    # same class count: plain non-strict load
    same = get_segmentation_module(_config(19), "train")
    ck.load_checkpoint({"paths": {"ckpt_model_path": path}, "classes": _classes(19, zero=(2,))}, same)
    assert all(torch.equal(a, b) for a, b in zip(same.state_dict().values(), src.state_dict().values()))
    # 19 -> 13 classes: head tensors become zeros of the first 13 rows, criterion.weight the config's weights
    cfg13 = _config(13)
    dst = get_segmentation_module(cfg13, "train")
    ck.load_checkpoint({"paths": {"ckpt_model_path": path}, "classes": cfg13["classes"]}, dst)
    sd, ssd = dst.state_dict(), src.state_dict()
    assert sd["model.seg_model.segmentation_head.0.weight"].shape == (13, 16, 3, 3)
    assert float(sd["model.seg_model.segmentation_head.0.weight"].abs().sum()) == 0.0
    assert float(sd["model.seg_model.segmentation_head.0.bias"].abs().sum()) == 0.0
    assert sd["criterion.weight"].tolist() == [cfg13["classes"][i][0] for i in cfg13["classes"]]
    k = "model.seg_model.encoder.layer3.2.conv1.weight"
    assert torch.equal(sd[k], ssd[k])
    # invalid paths / extensions
    with pytest.raises(SystemExit):
        ck.load_checkpoint({"paths": {"ckpt_model_path": str(tmp_path / "nope.ckpt")}, "classes": cfg13["classes"]}, dst, exit_on_fail=True)
    bad = str(tmp_path / "weights.bin")
    torch.save({}, bad)
    with pytest.raises(SystemExit):
        ck.load_checkpoint({"paths": {"ckpt_model_path": bad}, "classes": cfg13["classes"]}, dst, exit_on_fail=True)
    ck.load_checkpoint({"paths": {"ckpt_model_path": bad}, "classes": cfg13["classes"]}, dst)  # reports, returns


def test_resume_restores_optimizer_and_scheduler(tmp_path):
    from flair_amd import checkpoint as ck
    from flair_amd.tasks_utils import get_segmentation_module
    a = get_segmentation_module(_config(13), "train")
    for _ in range(12):
        a.scheduler.step(1.0)  # plateau: patience 10 -> one halving
    assert a.optimizer.param_groups[0]["lr"] == pytest.approx(0.01)
    path = str(tmp_path / "resume.ckpt")
    ck.save_checkpoint(path, a, a.optimizer, a.scheduler, epoch=12, global_step=48)
    b = get_segmentation_module(_config(13), "train")
    assert ck.resume(path, b, b.optimizer, b.scheduler) == (12, 48)
    assert b.optimizer.param_groups[0]["lr"] == pytest.approx(0.01)
    assert b.scheduler.state_dict()["cooldown_counter"] == a.scheduler.state_dict()["cooldown_counter"]
    assert all(torch.equal(x, y) for x, y in zip(a.state_dict().values(), b.state_dict().values()))
    assert os.path.getsize(path) > 90e6  # 24.4 M fp32 parameters


def test_metric_state_stays_out_of_checkpoints(tmp_path):
    """torchmetrics states are non-persistent, so the reference's .ckpt files hold the model and criterion.weight only;
    a checkpoint saved after setup('fit') must resume strictly into a fresh module and carry no metric keys."""
    import torch.nn as nn
    import flair_amd
    from flair_amd import checkpoint as ck
    C = 13

    def make():
        model = flair_amd.FLAIR_ModelFactory(_config(C), compute_dtype="f32")
        task = flair_amd.segmentation_task_training(model=model, class_infos=_classes(C), optimizer=None, scheduler=None,
                                                    criterion=nn.CrossEntropyLoss(weight=torch.ones(C)), use_metadata=False)
        return task

    a = make()
    a.setup("fit")
    keys = set(a.state_dict())
    want = {"model.seg_model." + k for k in om.seeded_model(5, C, seed=1).state_dict()} | {"criterion.weight"}
    assert keys == want, sorted(keys ^ want)[:8]
    path = str(tmp_path / "after_setup.ckpt")
    ck.save_checkpoint(path, a, epoch=1, global_step=7)
    b = make()
    ck.resume(path, b)          # strict=True inside: fails on any unexpected / missing key
    b.setup("fit")
    ck.resume(path, b)          # and the reverse order: a reference-style .ckpt strict-loaded after setup()
