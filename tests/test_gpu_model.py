"""Whole-path parity on the GPU: the HIP U-Net/ResNet34 behind the smp surface vs the CPU oracle
(oracle/unet_resnet34.py + oracle/seg_step.py) and vs the golden vectors made from the reference's own
model.py / task_module.py (tests/golden/make_golden.py).

Tolerances (north_star): fp32 mode logits within 1e-3 absolute; argmax masks bit-exact wherever the
oracle's top-2 probability gap exceeds 1e-5 (summation order differs between MKL-DNN and the MFMA
fp32 FMA chain, so exact ties may flip) with the flip count reported and bounded; bf16 mode is a
throughput mode judged by loss / mIoU closeness, never bit-exactness.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

CLASSES19 = {1: [1, 'building'], 2: [1, 'pervious surface'], 3: [1, 'impervious surface'], 4: [1, 'bare soil'],
             5: [1, 'water'], 6: [1, 'coniferous'], 7: [1, 'deciduous'], 8: [1, 'brushwood'], 9: [1, 'vineyard'],
             10: [1, 'herbaceous vegetation'], 11: [1, 'agricultural land'], 12: [1, 'plowed land'],
             13: [1, 'swimming_pool'], 14: [1, 'snow'], 15: [0, 'clear cut'], 16: [0, 'mixed'], 17: [0, 'ligneous'],
             18: [1, 'greenhouse'], 19: [0, 'other']}


def _config(n_ch, classes, use_metadata=False):
    return {"model_framework": {"model_provider": "SegmentationModelsPytorch",
                                "SegmentationModelsPytorch": {"encoder_decoder": "resnet34_unet", "encoder_weights": None}},
            "use_metadata": use_metadata, "channels": list(range(1, n_ch + 1)), "classes": classes}


def _pair(in_ch, classes, seed, dev, dtype="f32"):
    """Oracle model (CPU) and the HIP model loaded with the same seeded weights."""
    import flair_amd
    from oracle import unet_resnet34 as om
    ref = om.seeded_model(in_ch, classes, seed)
    hip = flair_amd.create_model("unet", "resnet34", encoder_weights=None, in_channels=in_ch, classes=classes, compute_dtype=dtype)
    hip.load_state_dict(ref.state_dict(), strict=True)
    return ref, hip.to(dev)


def test_golden_train_step_fp32(dev, golden_dir):
    """One training step (forward, weighted CE, backward, SGD lr .02) against the vector produced by the
    reference's FLAIR_ModelFactory + segmentation_task_training.step on the oracle model."""
    import flair_amd
    from oracle import unet_resnet34 as om
    g = np.load(os.path.join(golden_dir, "step_c13_b2_64.npz"))
    C = 13
    classes = {k: CLASSES19[k] for k in range(1, C + 1)}
    torch.manual_seed(int(g["seed"]))
    ref = om.create_model("unet", "resnet34", in_channels=5, classes=C)  # same RNG stream as make_golden.py
    model = flair_amd.FLAIR_ModelFactory(_config(5, classes), compute_dtype="f32")
    model.seg_model.load_state_dict(ref.state_dict(), strict=True)
    model = model.to(dev)
    assert sum(p.numel() for p in model.seg_model.parameters()) == int(g["n_params"]) == 24_444_381
    img = torch.from_numpy(g["img"]).to(dev)
    lab = torch.from_numpy(g["lab"]).long()
    msk = torch.stack([(lab == i) for i in range(C)], 1).float().to(dev)
    weight = torch.from_numpy(g["weight"])
    task = flair_amd.segmentation_task_training(model=model, class_infos=classes, criterion=nn.CrossEntropyLoss(weight=weight),
                                                optimizer=None, use_metadata=False, scheduler=None).to(dev)
    task.train()
    logits = task.forward(img, "")
    loss, preds, targets = flair_amd.task_module.fused_step(logits, msk, task.criterion)
    lg = logits.detach().cpu()
    assert (lg - torch.from_numpy(g["logits"])).abs().max() < 1e-3
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    assert torch.equal(targets.cpu().to(torch.uint8), torch.from_numpy(g["targets"]))
    from oracle import parity
    # masks: the reference's own predictions; zero mismatches wherever the top-2 probability gap exceeds 1e-5 (oracle/parity.py)
    parity.assert_mask_parity("golden_train_step_c13_b2_64", g["preds"].reshape(2, 64, 64), preds.cpu().view(2, 64, 64).numpy(),
                              parity.top2_gap(g["logits"]), logits_ref=g["logits"], logits_hip=lg.numpy(),
                              labels=g["targets"].reshape(2, 64, 64), num_classes=C)
    loss.backward()
    named = dict(model.seg_model.named_parameters())
    for k, v in zip(g["grad_keys"], g["grad_abs_sum"]):
        got = float(named[str(k)].grad.double().abs().sum())
        # |.|-sum checksums of gradients whose fp32 conditioning is ~1e-2 (see _assert_as_good_as_cpu_fp32)
        assert abs(got - v) <= 1e-2 * abs(v) + 1e-6, (k, got, v)
    # element level (round 3): 48 seeded elements of every gradient tensor of the REFERENCE's own step
    # (tests/golden/make_golden_grads.py), error relative to the tensor's max |.| (relative max-norm over the sample).
    # Two fp32 runs of this graph sit ~1e-2 apart where a pre-activation lies on a ReLU edge (see above): the bounds are the
    # measured distribution (profiles/r3_parity.json: median 2.3e-3, 90th percentile 3.7e-3, worst tensor 5.0e-3) x 2-4.
    gg = np.load(os.path.join(golden_dir, "step_c13_b2_64_grads.npz"))
    rel = {}
    for k, amax, idx, val in zip(gg["keys"], gg["abs_max"], gg["index"], gg["value"]):
        ok = idx >= 0
        got = named[str(k)].grad.detach().flatten().cpu().numpy()[idx[ok]]
        rel[str(k)] = float(np.abs(got.astype(np.float64) - val[ok]).max() / (amax + 1e-30))
    srt = sorted(rel.values())
    entry = {"test": "golden_step_gradient_elements", "tensors": len(srt), "samples_per_tensor": int(gg["index"].shape[1]),
             "median": srt[len(srt) // 2], "p90": srt[int(0.9 * len(srt))], "max": srt[-1], "worst_tensor": max(rel, key=rel.get)}
    parity.record(entry)
    assert entry["median"] <= 5e-3 and entry["p90"] <= 8e-3 and entry["max"] <= 2e-2, entry
    with torch.no_grad():
        task.forward(img, "")  # make_golden.py ran a second training-mode forward: running stats move twice
    torch.optim.SGD(model.parameters(), lr=float(g["lr"])).step()
    sd = model.seg_model.state_dict()
    for k, v in zip(g["post_keys"], g["post_abs_sum"]):
        got = float(sd[str(k)].double().abs().sum())
        assert abs(got - v) <= 1e-4 * abs(v) + 1e-5, (k, got, v)  # zero-initialised biases move by lr*grad only


def _grad_errs(params, truth):
    return {k: float((p.grad.detach().cpu().double() - truth[k].grad).abs().max() / (truth[k].grad.abs().max() + 1e-12))
            for k, p in params}


def _assert_as_good_as_cpu_fp32(e_hip, e_cpu, name="grad_vs_fp64", mult=(1.5, 2.0, 3.0)):
    """Gradients through 46 training-mode BatchNorms + ReLUs are ill-conditioned in fp32: a pre-activation
    within rounding of 0 flips its ReLU mask, so torch-CPU fp32 itself sits ~1e-2 (relative, max-norm) from
    an fp64 run.  The bar for the HIP fp32 path is therefore 'as close to fp64 as the reference arithmetic':
    per-tensor relative max-norm errors against the fp64 oracle, HIP vs torch-CPU fp32, compared at the median, the
    90th percentile and the maximum.  The measured ratios are logged (profiles/r2_parity.json); the multipliers are
    those measurements plus margin for the run-to-run spread of WHICH pre-activations sit on a ReLU edge: whole-model
    path on 2x5x160x128 measured 1.13x / 1.20x / 1.00x (bounds 1.5 / 2.0 / 3.0), split path on 1x5x512x512 1.57x / 2.06x /
    1.15x (bounds 2.0 / 2.5 / 3.0).  Round 3 (scripts/debug_split_grad.py, profiles/r3_split_grad.txt): the difference is the
    BATCH, not the path — on the same 1x512x512 batch and weights the whole-model path reads 1.23x / 1.69x / 1.14x and the split
    path 1.37x / 1.61x / 1.13x, on the same 2x160x128 batch 0.83x / 0.60x / 0.97x and 0.85x / 0.59x / 0.97x (a layout change of
    fp32 data is exact; with 262 144 pixels per channel in ONE image the BatchNorm-backward sums cancel harder, and torch's
    blocked CPU reductions happen to lose less there than the tile-ordered sums here; with two smaller images it is the other
    way round)."""
    from oracle import parity
    q = lambda d, f: sorted(d.values())[min(len(d) - 1, int(f * len(d)))]
    entry = {"test": name, "tensors": len(e_hip),
             "median_hip": q(e_hip, 0.5), "median_cpu_fp32": q(e_cpu, 0.5),
             "p90_hip": q(e_hip, 0.9), "p90_cpu_fp32": q(e_cpu, 0.9),
             "max_hip": max(e_hip.values()), "max_cpu_fp32": max(e_cpu.values())}
    parity.record(entry)
    entry["bounds"] = list(mult)
    assert q(e_hip, 0.5) <= mult[0] * q(e_cpu, 0.5) + 1e-4, entry
    assert q(e_hip, 0.9) <= mult[1] * q(e_cpu, 0.9) + 1e-3, entry
    # the maximum is a single-tensor outlier statistic (one flipped ReLU in a small layer): loose bound only
    assert max(e_hip.values()) <= mult[2] * max(e_cpu.values()) + 2e-2, entry


def test_train_step_matches_oracle_elementwise_fp32(dev):
    """Element-wise gradient parity on a 2x5x160x128 batch: HIP fp32 and torch-CPU fp32 both measured against
    the oracle run in fp64; logits, BN running statistics and the SGD update compared directly."""
    import copy
    from oracle import seg_step
    ref, hip = _pair(5, 13, 123, dev)
    ref64 = copy.deepcopy(ref).double()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 5, 160, 128, generator=g)
    lab = torch.randint(0, 13, (2, 160, 128), generator=g)
    ref.train(); hip.train(); ref64.train()
    l64 = ref64(x.double())
    nn.functional.cross_entropy(l64, lab).backward()
    lr_ = ref(x)
    nn.functional.cross_entropy(lr_, lab).backward()
    lh = hip(x.to(dev))
    nn.functional.cross_entropy(lh, lab.to(dev)).backward()
    assert (lh.detach().cpu() - lr_.detach()).abs().max() < 1e-3
    assert (lh.detach().cpu().double() - l64.detach()).abs().max() < 1e-3
    p64 = dict(ref64.named_parameters())
    _assert_as_good_as_cpu_fp32(_grad_errs(hip.named_parameters(), p64), _grad_errs(ref.named_parameters(), p64), "grad_vs_fp64_unet_2x5x160x128")
    for k, b in hip.named_buffers():
        rb = dict(ref.named_buffers())[k]
        assert torch.allclose(b.cpu().float(), rb.float(), rtol=1e-4, atol=1e-5), k
    before = {k: p.detach().clone() for k, p in hip.named_parameters()}
    torch.optim.SGD(hip.parameters(), lr=0.02).step()
    for k, p in hip.named_parameters():  # w <- w - lr*g exactly (tasks_utils.py:95: no momentum, no decay)
        assert torch.allclose(p.detach(), before[k] - 0.02 * p.grad, rtol=1e-6, atol=1e-9), k


def test_golden_predict_512_fp32(dev, golden_dir):
    """predict_step on one 512x512 tile in eval mode (running statistics after two training forwards)."""
    import flair_amd
    from oracle import unet_resnet34 as om
    gs = np.load(os.path.join(golden_dir, "step_c13_b2_64.npz"))
    gp = np.load(os.path.join(golden_dir, "predict_c13_512.npz"))
    torch.manual_seed(int(gs["seed"]))
    ref = om.create_model("unet", "resnet34", in_channels=5, classes=13)
    # replay make_golden.py on the oracle (two training forwards + one SGD step) to rebuild the state the
    # golden tile was predicted with — the golden file stores outputs only, not 93 MB of weights
    from oracle import seg_step
    img = torch.from_numpy(gs["img"])
    lab = torch.from_numpy(gs["lab"]).long()
    msk = torch.stack([(lab == i) for i in range(13)], 1).float()
    ref.train()
    loss, _, _ = seg_step.step_torch(ref(img), msk, torch.from_numpy(gs["weight"]))
    ref(img)
    loss.backward()
    seg_step.sgd_step_(ref.parameters(), 0.02)
    hip = flair_amd.create_model("unet", "resnet34", encoder_weights=None, in_channels=5, classes=13, compute_dtype="f32")
    hip.load_state_dict(ref.state_dict())
    hip = hip.to(dev)
    task = flair_amd.segmentation_task_predict(model=lambda im, met: hip(im), num_classes=13)
    hip.eval()
    tile = torch.randn(1, 5, 512, 512, generator=torch.Generator().manual_seed(int(gp["tile_seed"])))
    with torch.no_grad():
        out = task.predict_step({"img": tile.to(dev), "id": ["x"]}, 0)
        lg = hip(tile.to(dev)).cpu()
    assert sorted(out.keys()) == ["id", "img", "preds"] and out["preds"].dtype == torch.int64
    scale = max(1.0, float(np.abs(gp["logits_crop"]).max()))  # eval-mode logits of a random net reach ~13
    assert np.abs(lg[0, :, 240:272, 240:272].numpy() - gp["logits_crop"]).max() < 1e-3 * scale
    assert np.abs(lg.double().mean(dim=(0, 2, 3)).numpy() - gp["logits_mean"]).max() < 1e-3
    from oracle import parity
    ref.eval()
    with torch.no_grad():
        lg_ref = ref(tile).numpy()
    # the golden mask (made by the reference's predict_step on the oracle) against the HIP mask, gap from the oracle's logits
    parity.assert_mask_parity("golden_predict_c13_512", gp["preds"].reshape(1, 512, 512),
                              out["preds"].cpu().numpy().reshape(1, 512, 512), parity.top2_gap(lg_ref),
                              logits_ref=lg_ref, logits_hip=lg.numpy())


def test_golden_metadata_path_fp32(dev, golden_dir):
    """encoder -> metadata add -> decoder -> head (model.py:57-62) in eval mode, 19 classes, 512x512."""
    import flair_amd
    from oracle import unet_resnet34 as om
    gm = np.load(os.path.join(golden_dir, "metadata_c19_512.npz"))
    torch.manual_seed(2023)
    # make_golden.py: FLAIR_ModelFactory.__init__ builds MetadataMLP first, then the U-Net, under seed 2023
    enc_ref = flair_amd.MetadataMLP()
    ref = om.create_model("unet", "resnet34", in_channels=5, classes=19)
    model = flair_amd.FLAIR_ModelFactory(_config(5, CLASSES19, use_metadata=True), compute_dtype="f32")
    model.enc.load_state_dict(enc_ref.state_dict())
    model.seg_model.load_state_dict(ref.state_dict())
    model = model.to(dev).eval()
    g = torch.Generator().manual_seed(int(gm["tile_seed"]))
    tile = torch.randn(1, 5, 512, 512, generator=g)
    mtd = torch.from_numpy(gm["mtd"])
    with torch.no_grad():
        assert np.abs(model.enc(mtd.to(dev)).cpu().numpy() - gm["x_enc"]).max() < 1e-5
        lg = model(tile.to(dev), mtd.to(dev)).cpu()
    scale = np.abs(gm["logits_crop"]).max()
    assert np.abs(lg[0, :, 100:132, 300:332].numpy() - gm["logits_crop"]).max() < 1e-3 * max(1.0, scale)
    assert np.abs(lg.double().mean(dim=(0, 2, 3)).numpy() - gm["logits_mean"]).max() < 1e-3


def test_split_path_training_matches_fused(dev):
    """encoder/decoder/head called separately (with a metadata add in between) give the same gradients
    as autograd through the oracle."""
    import flair_amd
    from oracle import unet_resnet34 as om
    ref = om.seeded_model(5, 13, 77)
    model = flair_amd.FLAIR_ModelFactory(_config(5, {k: CLASSES19[k] for k in range(1, 14)}, use_metadata=True), compute_dtype="f32")
    model.seg_model.load_state_dict(ref.state_dict())
    model = model.to(dev).train()
    model.enc.eval()  # dropout off for parity (quirk Q5)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(1, 5, 512, 512, generator=g)
    mtd = torch.rand(1, 45, generator=g)
    lab = torch.randint(0, 13, (1, 512, 512), generator=g)
    out = model(x.to(dev), mtd.to(dev))
    nn.functional.cross_entropy(out, lab.to(dev)).backward()
    # oracle with the reference's own fusion arithmetic (model.py:57-62)
    import copy
    enc_cpu = copy.deepcopy(model.enc).cpu()
    ref.train()
    feats = ref.encoder(x)
    x_enc = enc_cpu(mtd).unsqueeze(1).unsqueeze(-1).repeat(1, 512, 1, 16)
    feats[-1] = torch.add(feats[-1], x_enc)
    o = ref.segmentation_head(ref.decoder(*feats))
    nn.functional.cross_entropy(o, lab).backward()
    assert (out.detach().cpu() - o.detach()).abs().max() < 1e-3
    # fp64 ground truth for the same computation
    ref64, enc64 = copy.deepcopy(ref).double(), copy.deepcopy(enc_cpu).double()
    ref64.zero_grad(); enc64.zero_grad()
    f64 = ref64.encoder(x.double())
    f64[-1] = torch.add(f64[-1], enc64(mtd.double()).unsqueeze(1).unsqueeze(-1).repeat(1, 512, 1, 16))
    nn.functional.cross_entropy(ref64.segmentation_head(ref64.decoder(*f64)), lab).backward()
    p64 = dict(ref64.named_parameters())
    _assert_as_good_as_cpu_fp32(_grad_errs(model.seg_model.named_parameters(), p64), _grad_errs(ref.named_parameters(), p64), "grad_vs_fp64_split_path_unet", (2.0, 2.5, 3.0))
    e64 = dict(enc64.named_parameters())
    _assert_as_good_as_cpu_fp32(_grad_errs(model.enc.named_parameters(), e64), _grad_errs(enc_cpu.named_parameters(), e64), "grad_vs_fp64_split_path_mlp", (2.0, 2.5, 3.0))


def test_bf16_throughput_mode_tracks_fp32(dev):
    """bf16 mode: same step on the same weights stays close in loss and mIoU (not bit-exact by design)."""
    import flair_amd
    from oracle import seg_step
    ref, hip32 = _pair(5, 13, 31, dev, "f32")
    _, hip16 = _pair(5, 13, 31, dev, "bf16")
    g = torch.Generator().manual_seed(2)
    x = torch.randn(4, 5, 128, 128, generator=g).to(dev)
    lab = torch.randint(0, 13, (4, 128, 128), generator=g).to(dev)
    res = {}
    for name, m in (("f32", hip32), ("bf16", hip16)):
        tr = flair_amd.SegTrainer(m.train(), lr=0.02)
        losses = [tr.train_step(x, lab.to(torch.uint8)).item() for _ in range(3)]
        cm = tr.confmat.cpu().numpy()
        res[name] = (losses, seg_step.jaccard_from_confmat(cm, "weighted"), tr.predict(x).cpu())
    l32, l16 = res["f32"][0], res["bf16"][0]
    assert all(abs(a - b) < 0.05 * abs(a) for a, b in zip(l32, l16)), (l32, l16)
    assert l32[-1] < l32[0]  # SGD makes progress on a fixed batch
    assert abs(res["f32"][1] - res["bf16"][1]) < 0.02
    agree = float((res["f32"][2] == res["bf16"][2]).float().mean())
    assert agree > 0.9, agree


@pytest.mark.parametrize("C", [13, 19])
def test_trainer_matches_autograd_path_fp32(dev, C):
    """SegTrainer.train_step (fused C path) == autograd path + torch SGD, bit for bit on the weights.  19 classes: the head's
    NHWC rows are 24 wide (round 3: the fused head refused that row length, so the trainer never ran BASELINE config 4's class
    count)."""
    import flair_amd
    _, a = _pair(5, C, 5, dev)
    _, b = _pair(5, C, 5, dev)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 5, 64, 64, generator=g).to(dev)
    lab = torch.randint(0, C, (2, 64, 64), generator=g).to(dev)
    w = torch.linspace(0.5, 2, C)
    tr = flair_amd.SegTrainer(a.train(), lr=0.02, class_weight=w)
    la = tr.train_step(x, lab.to(torch.uint8))
    crit = flair_amd.FusedCrossEntropyLoss(weight=w).to(dev)
    b.train()
    lb = crit(b(x), lab)
    lb.backward()
    torch.optim.SGD(b.parameters(), lr=0.02).step()
    assert abs(la.item() - lb.item()) < 1e-6
    assert torch.allclose(a.flat_parameters(), b.flat_parameters(), rtol=0, atol=1e-7)
    assert torch.equal(crit.last_preds.to(torch.uint8), tr._preds)


def test_errors_and_contract(dev):
    import flair_amd
    from flair_amd._lib import FlairHipError
    m = flair_amd.create_model("unet", "resnet34", encoder_weights=None, in_channels=5, classes=13)
    with pytest.raises(FlairHipError):
        m(torch.zeros(1, 5, 64, 64))  # host tensors are refused: no CPU fallback
    m = m.to(dev)
    with pytest.raises(RuntimeError, match="divisible by 32"):
        m(torch.zeros(1, 5, 70, 64, device=dev))
    with pytest.raises(KeyError):
        flair_amd.create_model("fpn", "resnet34")
    with pytest.raises(KeyError):
        flair_amd.create_model("unet", "resnet50")
    assert any(isinstance(x, nn.BatchNorm2d) for x in m.modules())
    feats = m.eval().encoder(torch.zeros(1, 5, 64, 64, device=dev))
    assert [tuple(f.shape[1:]) for f in feats] == [(5, 64, 64), (64, 32, 32), (64, 16, 16), (128, 8, 8), (256, 4, 4), (512, 2, 2)]


@pytest.mark.parametrize("shape", [(3, 96, 160), (2, 64, 96), (2, 224, 64)])
def test_bf16_tracks_fp32_like_torch_autocast_on_ragged_shapes(dev, shape):
    """Odd batch sizes and non-square images send some layers to the tile kernels and others to the gather-form /
    generic fall-backs (and the stem, parity-class and lazy-BN paths in or out of their size windows).  bf16 gradients
    through 46 training-mode BatchNorms are noisy by nature (the BN backward subtracts the two dominant components of
    dz): torch's own CPU autocast(bf16) run of the oracle has a per-tensor cosine of only ~0.65 to its fp32 run on
    random data.  The bar for the HIP bf16 mode is therefore: loss and per-stage gradient norms close to HIP fp32, and a
    gradient direction at least as close to fp32 as torch autocast's."""
    import copy
    import flair_amd
    B, H, W = shape
    ref, hip32 = _pair(5, 13, 77, dev, "f32")
    _, hip16 = _pair(5, 13, 77, dev, "bf16")
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn(B, 5, H, W, generator=g)
    lab = torch.randint(0, 13, (B, H, W), generator=g)
    out = {}
    for name, m in (("f32", hip32), ("bf16", hip16)):
        tr = flair_amd.SegTrainer(m.train(), lr=0.0)   # gradients only
        loss = tr.train_step(x.to(dev), lab.to(torch.uint8).to(dev)).item()
        names = [n for n, _ in m.named_parameters()]
        out[name] = (loss, [tr.grads[b:e].double().norm().item() for b, e in m.stage_ranges()],
                     dict(zip(names, [v.detach().cpu().double().flatten() for v in m._grad_views(tr.grads)])))
    (l32, n32, g32), (l16, n16, g16) = out["f32"], out["bf16"]
    assert abs(l32 - l16) < 0.03 * abs(l32), (l32, l16)
    for stage, (a, b) in enumerate(zip(n32, n16)):
        assert abs(a - b) <= 0.25 * a + 1e-6, (stage, a, b)

    def cpu_grads(amp):
        m = copy.deepcopy(ref).train()
        if amp:
            with torch.autocast(device_type="cpu", dtype=torch.bfloat16):
                logits = m(x)
            nn.functional.cross_entropy(logits.float(), lab).backward()
        else:
            nn.functional.cross_entropy(m(x), lab).backward()
        return {k: p.grad.double().flatten() for k, p in m.named_parameters()}

    c32, c16 = cpu_grads(False), cpu_grads(True)
    cos = lambda a, b: float(nn.functional.cosine_similarity(a, b, dim=0))
    med = lambda v: sorted(v)[len(v) // 2]
    hip_cos = med([cos(g32[k], g16[k]) for k in g32])
    amp_cos = med([cos(c32[k], c16[k]) for k in c32])
    assert hip_cos >= amp_cos - 0.05, (hip_cos, amp_cos)
    # and the fp32 HIP gradients point where torch-CPU fp32's do
    assert med([cos(g32[k], c32[k]) for k in g32]) > 0.999


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_nhwc_logits_head_paths_equal_the_nchw_ones(dev, dtype):
    """The fused trainer never materialises fp32 NCHW logits (flair_ce_head_nhwc / flair_softmax_argmax_nhwc over the
    head's NHWC output): loss, predictions, confusion matrix and masks must equal the NCHW entry points bit for bit."""
    import flair_amd
    from flair_amd import ops
    _, hip = _pair(5, 13, 9, dev, dtype)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 5, 64, 96, generator=g).to(dev)
    lab = torch.randint(0, 15, (2, 64, 96), generator=g).to(torch.uint8).to(dev)   # labels >= 13 carry no weight
    w = torch.linspace(0.5, 1.5, 13)
    w[4] = 0.0
    tr = flair_amd.SegTrainer(hip.train(), lr=0.0, class_weight=w)
    loss_t = tr.train_step(x, lab).item()
    preds_t, cm_t = tr._preds.clone(), tr.confmat.clone()
    logits = hip._c_forward(x, training=True)            # same weights (lr = 0), same batch statistics
    cm = torch.zeros(13, 13, dtype=torch.int64, device=dev)
    loss, _, preds, _ = ops.ce_head(logits, lab, w.to(dev), want_dlogits=False, want_preds="u8", confmat=cm)
    assert abs(loss_t - loss.item()) <= 2e-6 * abs(loss_t)   # same terms, different partial-sum grouping
    assert torch.equal(preds_t, preds) and torch.equal(cm_t, cm)
    hip.eval()
    with torch.no_grad():
        assert torch.equal(tr.predict(x), ops.softmax_argmax(hip(x).float().contiguous(), want="u8"))


def test_config4_train_step_metadata_weighted_d4(dev):
    """19 classes with the YAML weights (four of them zero), metadata fusion through MetadataMLP
    (dropout off: eval() on the MLP only, so the comparison does not depend on an RNG stream), D4 augmentation and
    'custom' normalisation on the device feed — one training step against the oracle on the same batch."""
    import flair_amd
    from flair_amd.data_feed import TileFeed, pack_d4
    from oracle import unet_resnet34 as om
    C, B = 19, 2
    cfg = _config(5, CLASSES19, use_metadata=True)
    torch.manual_seed(3)
    hip = flair_amd.FLAIR_ModelFactory(cfg, compute_dtype="f32")
    ref = om.seeded_model(5, C, 11)
    hip.seg_model.load_state_dict(ref.state_dict(), strict=True)
    mlp = flair_amd.MetadataMLP()
    mlp.load_state_dict(hip.enc.state_dict())
    hip = hip.to(dev).train()
    hip.enc.eval()
    ref.train(); mlp.eval()
    g = torch.Generator().manual_seed(21)
    img_u8 = torch.randint(0, 256, (B, 5, 512, 512), dtype=torch.uint8, generator=g)
    raw = torch.randint(0, C + 2, (B, 512, 512), dtype=torch.uint8, generator=g)
    mtd = torch.rand(B, 45, generator=g)
    feed = TileFeed([1, 2, 3, 4, 5], C, "custom", [105.08, 110.87, 101.82, 106.38, 53.26], [52.17, 45.38, 44, 39.69, 79.3])
    batch = feed(img_u8.to(dev), raw.to(dev), d4=torch.tensor([pack_d4(1, 0, 3), pack_d4(0, 1, 2)], dtype=torch.uint8), mtd=mtd)
    weight = torch.tensor([CLASSES19[k][0] for k in CLASSES19], dtype=torch.float32)
    crit = flair_amd.FusedCrossEntropyLoss(weight=weight).to(dev)
    loss = crit(hip(batch["img"], batch["mtd"]), batch["msk"])
    loss.backward()
    # oracle: the reference's forward (model.py:57-62) on the same float batch
    x, lab = batch["img"].cpu(), batch["msk"].cpu().long()
    feats = ref.encoder(x)
    enc = mlp(mtd)
    feats[-1] = torch.add(feats[-1], enc.unsqueeze(1).unsqueeze(-1).repeat(1, 512, 1, 16))
    logits = ref.segmentation_head(ref.decoder(*feats))
    loss_ref = nn.functional.cross_entropy(logits, lab, weight=weight)
    loss_ref.backward()
    assert abs(loss.item() - loss_ref.item()) < 2e-4 * abs(loss_ref.item()), (loss.item(), loss_ref.item())
    cos = lambda a, b: float(nn.functional.cosine_similarity(a.double().flatten(), b.double().flatten(), dim=0))
    pr = dict(ref.named_parameters())
    cs = sorted(cos(p.grad.cpu(), pr[k].grad) for k, p in hip.seg_model.named_parameters())
    assert cs[len(cs) // 2] > 0.9999 and cs[0] > 0.99, (cs[0], cs[len(cs) // 2])
    # the metadata branch: gradient reaches the MLP through the row-vector add
    pm = dict(mlp.named_parameters())
    for k, p in hip.enc.named_parameters():
        assert p.grad is not None and cos(p.grad.cpu(), pm[k].grad) > 0.999, k


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_training_step_is_bit_reproducible(dev, dtype):
    """No float atomics anywhere and fixed-order two-stage reductions: the same step from the same weights gives the same
    bits — loss, every gradient, BN running statistics — also with the weight-gradient kernels on their own stream."""
    import flair_amd
    ref, a = _pair(5, 13, 5, dev, dtype)
    _, b = _pair(5, 13, 5, dev, dtype)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 5, 256, 256, generator=g).to(dev)
    lab = torch.randint(0, 13, (4, 256, 256), generator=g).to(torch.uint8).to(dev)
    out = []
    for m in (a, b, a):
        tr = flair_amd.SegTrainer(m.train(), lr=0.0)
        loss = tr.train_step(x, lab)
        torch.cuda.synchronize()
        out.append((loss.item(), tr.grads.clone(), m.flat_buffers().clone()))
    assert out[0][0] == out[1][0] and torch.equal(out[0][1], out[1][1])
    assert torch.equal(out[0][1], out[2][1])   # and again on the first model (lr = 0: weights unchanged)
    assert torch.equal(out[0][2], out[1][2])


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_step_does_not_see_the_previous_steps_partial_sums(dev, dtype):
    """The arena hands out the same addresses every step (partial sums, slabs, packed weights, gradient buffers): anything
    read before it is rewritten would show the PREVIOUS step's data.  With lr = 0 the gradients of batch Y must be
    bit-identical whether or not a different batch X went through the model first (and on a second model that never saw
    X).  (Written for an in-launch BatchNorm-backward finalize that was measured and dropped; kept as a
    guard for every two-stage reduction.)"""
    import flair_amd
    _, a = _pair(5, 13, 9, dev, dtype)
    _, b = _pair(5, 13, 9, dev, dtype)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 5, 128, 192, generator=g).to(dev) * 3.0
    y = torch.randn(4, 5, 128, 192, generator=g).to(dev)
    lx = torch.randint(0, 13, (4, 128, 192), generator=g).to(torch.uint8).to(dev)
    ly = torch.randint(0, 13, (4, 128, 192), generator=g).to(torch.uint8).to(dev)
    ta = flair_amd.SegTrainer(a.train(), lr=0.0)
    tb = flair_amd.SegTrainer(b.train(), lr=0.0)
    for _ in range(3):
        ta.train_step(x, lx)      # model a has X's partial sums in its arena
    ta.train_step(y, ly)
    tb.train_step(y, ly)
    torch.cuda.synchronize()
    assert ta.loss.item() == tb.loss.item()
    assert torch.equal(ta.grads, tb.grads)
    for _ in range(2):            # and alternating, in one arena
        ta.train_step(x, lx)
        ta.train_step(y, ly)
        torch.cuda.synchronize()
        assert torch.equal(ta.grads, tb.grads)


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_full_size_step_properties(dev, dtype):
    """BASELINE configuration (32 x 5 x 512 x 512, 13 classes), checked through size-independent properties:
    a batch made of the same 16 tiles twice gives pairwise identical masks (every per-pixel reduction has a fixed order
    that does not depend on where the tile sits in the batch), the confusion matrix counts every pixel exactly once in
    the row of its label and the column of its prediction, and the mean loss equals that of the 16-tile batch.  In fp32
    the BatchNorm running statistics and the gradient of the doubled batch also equal the 16-tile ones up to summation
    grouping; in bf16 ulp-level differences of the BN coefficients are amplified through 46 BatchNorm backward passes
    (cosine ~0.95), so only the exact properties are asserted there."""
    import flair_amd
    _, m = _pair(5, 13, 2022, dev, dtype)
    g = torch.Generator().manual_seed(3)
    x16 = torch.randn(16, 5, 512, 512, generator=g)
    lab16 = torch.randint(0, 13, (16, 512, 512), generator=g).to(torch.uint8)
    x, lab = torch.cat([x16, x16]).to(dev), torch.cat([lab16, lab16]).to(dev)
    tr = flair_amd.SegTrainer(m.train(), lr=0.0)
    loss32 = tr.train_step(x, lab).item()
    preds, cm, g32, buf32 = tr._preds.clone(), tr.confmat.clone(), tr.grads.clone(), m.flat_buffers().clone()
    assert torch.equal(preds[:16], preds[16:])
    assert int(cm.sum()) == 32 * 512 * 512
    assert torch.equal(cm.sum(1), torch.bincount(lab.flatten().long(), minlength=13))
    assert torch.equal(cm.sum(0), torch.bincount(preds.flatten().long(), minlength=13))
    del tr
    _, m2 = _pair(5, 13, 2022, dev, dtype)
    tr2 = flair_amd.SegTrainer(m2.train(), lr=0.0)
    loss16 = tr2.train_step(x[:16].contiguous(), lab[:16].contiguous()).item()
    assert abs(loss32 - loss16) < 1e-5 * abs(loss16)
    if dtype == "f32":
        assert (tr2._preds == preds[:16]).float().mean() > 0.999
        assert (buf32 - m2.flat_buffers()).abs().max() < 1e-4
        cos = torch.nn.functional.cosine_similarity(g32.double(), tr2.grads.double(), dim=0).item()
        assert cos > 0.999 and abs(float(g32.norm() / tr2.grads.norm()) - 1) < 5e-3, cos


# bf16 logit error against the fp32 oracle, relative to the logit scale max(1, max |logit|) of a freshly initialised network on
# noise tiles (47 layers, every activation rounded to 8 significant bits): the WORST of the 3.4 M logits of a 512x512 tile sits
# 0.12 of the scale off (0.76 of 6.6), the rms error is 1.6e-2 of it — measured, profiles/r3_parity.json; bounds = 1.6x those
BF16_REL_DLOGIT = 0.2
BF16_REL_RMS = 0.03


@pytest.mark.parametrize("B,classes", [(1, 13), (2, 19)])
def test_bf16_logits_and_masks_against_the_fp32_oracle_at_512(dev, B, classes):
    """The BENCHMARKED arithmetic (bf16 throughput mode) against the oracle itself at the tile size of BASELINE configs 2 and 4:
    training-mode forward (batch statistics, what the bench step runs) and predict (eval mode).  bf16 cannot be bit-exact, so
    the statement is the logit-error rule of oracle/parity.py: with delta = max |logit_bf16 - logit_oracle| <= BF16_REL_DLOGIT x
    logit scale, the HIP mask may differ from the oracle's ONLY where the oracle's top-2 logit gap is <= 2 delta — zero
    mismatches on every pixel the oracle decides by more than the bf16 logit error."""
    import flair_amd
    from oracle import parity
    ref, hip = _pair(5, classes, 41, dev, "bf16")
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, 5, 512, 512, generator=g)
    lab = torch.randint(0, classes, (B, 512, 512), generator=g).to(torch.uint8)
    tr = flair_amd.SegTrainer(hip.train(), lr=0.0)
    tr.train_step(x.to(dev), lab.to(dev))          # masks of the training-mode forward: tr._preds
    ref.train()
    with torch.no_grad():
        lg_ref = ref(x)
    hip2 = flair_amd.create_model("unet", "resnet34", encoder_weights=None, in_channels=5, classes=classes, compute_dtype="bf16")
    hip2.load_state_dict(_pair(5, classes, 41, dev, "bf16")[0].state_dict())
    hip2 = hip2.to(dev).train()
    with torch.no_grad():
        lg_hip = hip2(x.to(dev)).cpu()             # the same forward with fp32 NCHW logits out
    e1 = parity.assert_masks_within_logit_error(f"bf16_train_forward_{B}x512_c{classes}", lg_ref.numpy(), lg_hip.numpy(),
                                                tr._preds.cpu().numpy(), max_rel_dlogit=BF16_REL_DLOGIT, max_rel_rms=BF16_REL_RMS)
    # predict: eval mode on identical weights and running statistics (the oracle's, after its training forward above)
    hip2.load_state_dict(ref.state_dict())
    hip2.eval(); ref.eval()
    with torch.no_grad():
        pe_ref = ref(x)
        pe_hip = hip2(x.to(dev)).cpu()
    preds = flair_amd.SegTrainer(hip2, lr=0.0).predict(x.to(dev)).cpu().numpy()
    e2 = parity.assert_masks_within_logit_error(f"bf16_predict_{B}x512_c{classes}", pe_ref.numpy(), pe_hip.numpy(), preds,
                                                max_rel_dlogit=BF16_REL_DLOGIT, max_rel_rms=BF16_REL_RMS)
    print(f"bf16 vs oracle {B}x512 c{classes}: train fwd dlogit {e1['max_abs_dlogit']:.3e} / scale {e1['logit_scale']:.2f}, "
          f"{e1['flips']} flips (largest gap {e1['max_logit_gap_among_flipped']:.2e}); predict dlogit {e2['max_abs_dlogit']:.3e} / "
          f"scale {e2['logit_scale']:.2f}, {e2['flips']} flips")


def test_bf16_masks_against_hip_fp32_at_the_bench_batch(dev):
    """BASELINE config 2 itself (32 x 5 x 512 x 512, 13 classes): the bf16 step's masks against the fp32 parity mode's on the
    same weights and batch, under the same rule (the oracle needs minutes at this size; HIP fp32 is what the oracle tests pin).
    Replaces the 'agreement > 0.9' of round 2: every disagreement must lie where fp32's top-2 logit gap is within twice the
    measured bf16 logit error."""
    import flair_amd
    from oracle import parity
    g = torch.Generator().manual_seed(12)
    x = torch.randn(32, 5, 512, 512, generator=g).to(dev)
    lab = torch.randint(0, 13, (32, 512, 512), generator=g).to(torch.uint8).to(dev)
    out = {}
    for dt in ("f32", "bf16"):
        _, m = _pair(5, 13, 2022, dev, dt)
        tr = flair_amd.SegTrainer(m.train(), lr=0.0)
        tr.train_step(x, lab)
        masks = tr._preds.cpu().numpy()
        del tr
        _, m2 = _pair(5, 13, 2022, dev, dt)
        with torch.no_grad():
            out[dt] = (m2.to(dev).train()(x).cpu().numpy(), masks)
        del m, m2
        torch.cuda.empty_cache()
    # fp32's own masks are the argmax of its own logits (first maximum)
    assert np.array_equal(out["f32"][0].argmax(1), out["f32"][1])
    e = parity.assert_masks_within_logit_error("bf16_vs_hip_fp32_32x512_c13", out["f32"][0], out["bf16"][0], out["bf16"][1],
                                               max_rel_dlogit=BF16_REL_DLOGIT, max_rel_rms=BF16_REL_RMS)
    agree = 1.0 - e["flips"] / e["pixels"]
    print(f"bf16 vs fp32 at B=32: dlogit {e['max_abs_dlogit']:.3e} / scale {e['logit_scale']:.2f}, agreement {agree:.5f}")
    assert agree > 0.8   # (noise tiles through a random network: the top-2 logit gap of most pixels is below the bf16 error)


def test_eval_forward_between_training_forward_and_backward(dev):
    """Legal in PyTorch (validation / predict on the model between loss computation and backward): eval-mode forwards run
    on their own native executor and workspace, so the recorded training graph survives — gradients equal those of an
    undisturbed step bit for bit."""
    ref, hip = _pair(5, 13, 77, dev)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 5, 64, 64, generator=g).to(dev)
    lab = torch.randint(0, 13, (2, 64, 64), generator=g).to(dev)
    hip.train()

    def grads(disturb):
        for p in hip.parameters():
            p.grad = None
        hip.load_state_dict(ref.state_dict(), strict=True)   # same weights and running statistics for both runs
        loss = nn.functional.cross_entropy(hip(x), lab)
        if disturb:
            hip.eval()
            with torch.no_grad():
                hip(torch.flip(x, dims=[3]))                  # whole-model eval forward
                hip.predict(x[:1]) if hasattr(hip, "predict") else None
            hip.train()
        loss.backward()
        return [p.grad.clone() for p in hip.parameters()]

    a, b = grads(False), grads(True)
    assert all(torch.equal(u, v) for u, v in zip(a, b))


def test_checkpoint_files_drive_the_hip_model(dev, tmp_path):
    """f4 on the device (main.py:77-146, zone_detect/model.py:61-88): a Lightning-style .ckpt written from the ORACLE's
    weights -> flair_amd.checkpoint.{load_checkpoint, load_model, resume} -> HIP forward == oracle forward; the 19 -> 13
    class surgery zeroes the head (logits become exactly the zero bias) and leaves the trunk intact."""
    import flair_amd
    from flair_amd import checkpoint as ck
    from flair_amd.tasks_utils import get_segmentation_module
    from oracle import parity
    from oracle import unet_resnet34 as om
    C = 19
    ref = om.seeded_model(5, C, seed=31).eval()
    with torch.no_grad():   # non-trivial running statistics, as a trained checkpoint has
        for m in ref.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.uniform_(-0.2, 0.2, generator=torch.Generator().manual_seed(1))
                m.running_var.uniform_(0.5, 1.5, generator=torch.Generator().manual_seed(2))
    lsd = {"model.seg_model." + k: v for k, v in ref.state_dict().items()}
    lsd["criterion.weight"] = torch.ones(C)
    path = str(tmp_path / "trained.ckpt")
    torch.save({"state_dict": lsd, "epoch": 3, "global_step": 99, "optimizer_states": [], "lr_schedulers": []}, path)
    x = torch.randn(2, 5, 96, 64, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        want = ref(x)
    classes = {k: CLASSES19[k] for k in range(1, C + 1)}
    cfg = dict(_config(5, classes), use_weights=True, learning_rate=0.02, paths={"ckpt_model_path": path})

    def check(seg, name):
        seg = seg.to(dev).eval()
        with torch.no_grad():
            got = seg(x.to(dev)).cpu()
        assert float((got - want).abs().max()) < 1e-3 * max(1.0, float(want.abs().max()))
        parity.assert_mask_parity(name, want.argmax(1).numpy(), got.argmax(1).numpy(), parity.top2_gap(want.numpy()),
                                  logits_ref=want.numpy(), logits_hip=got.numpy())

    # (1) main.py's path: load_checkpoint into the training task, same class count
    task = get_segmentation_module(cfg, "train", compute_dtype="f32")
    ck.load_checkpoint(cfg, task)
    check(task.model.seg_model, "ckpt_load_checkpoint_c19")
    # (2) Trainer.fit(ckpt_path=...) -> resume (strict)
    task2 = get_segmentation_module(cfg, "train", compute_dtype="f32")
    assert ck.resume(path, task2) == (3, 99)
    check(task2.model.seg_model, "ckpt_resume_c19")
    # (3) zone_detect's load_model (prefix stripping, strict load into the bare segmentation model)
    zcfg = dict(cfg, model_weights=path)
    check(ck.load_model(zcfg, compute_dtype="f32"), "ckpt_zone_detect_load_model_c19")
    # (4) 19 -> 13 classes: head rows zeroed -> logits == 0 everywhere, encoder tensors untouched
    classes13 = {k: CLASSES19[k] for k in range(1, 14)}
    cfg13 = dict(_config(5, classes13), use_weights=True, learning_rate=0.02, paths={"ckpt_model_path": path})
    task13 = get_segmentation_module(cfg13, "train", compute_dtype="f32")
    ck.load_checkpoint(cfg13, task13)
    seg13 = task13.model.seg_model.to(dev).eval()
    with torch.no_grad():
        lg13 = seg13(x.to(dev)).cpu()
    assert lg13.shape == (2, 13, 96, 64) and float(lg13.abs().max()) == 0.0
    k = "encoder.layer2.1.conv2.weight"
    assert torch.equal(seg13.state_dict()[k].cpu(), ref.state_dict()[k])


def test_metadata_mlp_kernel_matches_reference_module(dev):
    """MetadataMLP (model.py:74-96) as one HIP kernel each way: eval mode against the plain nn.Sequential of the
    reference, training mode with SUPPLIED dropout masks against the same arithmetic in torch (outputs 1e-6, parameter
    gradients 1e-5 relative); state_dict keys are the reference's."""
    import flair_amd
    torch.manual_seed(5)
    mlp = flair_amd.MetadataMLP()
    assert list(mlp.state_dict()) == ["enc_mlp.0.weight", "enc_mlp.0.bias", "enc_mlp.3.weight", "enc_mlp.3.bias",
                                      "enc_mlp.6.weight", "enc_mlp.6.bias"]
    x = torch.randn(7, 45)
    mlp.eval()
    with torch.no_grad():
        want = mlp(x)                       # host tensors: the reference's module tree
    hip = flair_amd.MetadataMLP()
    hip.load_state_dict(mlp.state_dict())
    hip = hip.to(dev).eval()
    with torch.no_grad():
        got = hip(x.to(dev)).cpu()
    assert float((got - want).abs().max()) < 1e-6
    # training mode, masks supplied (Dropout sits BEFORE the ReLU in the reference: mask * pre-activation)
    hip.train()
    masks = [torch.bernoulli(torch.full((7, n), 0.6)) / 0.6 for n in (64, 32, 16)]
    out = hip(x.to(dev), masks=[m.to(dev) for m in masks])
    w = torch.randn(7, 16)
    (out * w.to(dev)).sum().backward()
    ref = [p.detach().clone().requires_grad_(True) for p in mlp.parameters()]
    h = x
    for i in range(3):
        h = torch.relu((h @ ref[2 * i].t() + ref[2 * i + 1]) * masks[i])
    assert float((out.detach().cpu() - h).abs().max()) < 1e-6
    (h * w).sum().backward()
    for p, r in zip(hip.parameters(), ref):
        assert float((p.grad.cpu() - r.grad).abs().max()) <= 1e-5 * max(1.0, float(r.grad.abs().max()))
    # any batch size, like the reference's nn.Sequential (the backward kernel takes 256 samples per launch: slices)
    xb = torch.randn(300, 45)
    hip.zero_grad()
    hip.eval()
    with torch.no_grad():
        assert float((hip(xb.to(dev)).cpu() - mlp(xb)).abs().max()) < 1e-6
    hip.train()
    mb = [torch.bernoulli(torch.full((300, n), 0.6)) / 0.6 for n in (64, 32, 16)]
    ob = hip(xb.to(dev), masks=[m.to(dev) for m in mb])
    ob.sum().backward()
    ref = [p.detach().clone().requires_grad_(True) for p in mlp.parameters()]
    h = xb
    for i in range(3):
        h = torch.relu((h @ ref[2 * i].t() + ref[2 * i + 1]) * mb[i])
    assert float((ob.detach().cpu() - h).abs().max()) < 1e-6
    h.sum().backward()
    for p, r in zip(hip.parameters(), ref):
        assert float((p.grad.cpu() - r.grad).abs().max()) <= 2e-5 * max(1.0, float(r.grad.abs().max()))


def test_constant_weight_reuse_in_eval_loops_is_invalidated_by_every_writer(dev):
    """Eval forwards with unchanged weights skip the weight pack and the BatchNorm-coefficient launches
    (flair_unet_reuse_constants).  Same outputs as a cold forward, and every way the weights can change between two
    eval forwards — torch in-place writes, load_state_dict, a training forward (running statistics), the fused
    trainer's native SGD — is noticed."""
    import flair_amd
    ref, hip = _pair(5, 13, 11, dev)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 5, 64, 64, generator=g).to(dev)
    y = torch.randn(2, 5, 64, 64, generator=g).to(dev)
    lab = torch.randint(0, 13, (2, 64, 64), generator=g).to(torch.uint8).to(dev)
    hip.eval()

    def cold(inp):   # a fresh model holding the same state: never reuses anything
        m = flair_amd.create_model("unet", "resnet34", encoder_weights=None, in_channels=5, classes=13, compute_dtype="f32")
        m.load_state_dict(hip.state_dict())
        m = m.to(dev).eval()
        with torch.no_grad():
            return m(inp)

    with torch.no_grad():
        a = hip(x)
        b = hip(y)            # second eval forward: constants reused
        assert torch.equal(b, cold(y)) and torch.equal(hip(x), a)
        # 1. torch in-place write to a parameter
        next(iter(hip.parameters())).mul_(1.5)
        assert torch.equal(hip(y), cold(y))
        # 2. load_state_dict
        hip.load_state_dict(ref.state_dict())
        assert torch.equal(hip(y), cold(y))
    # 3. a training forward moves the running statistics natively
    hip.train()
    with torch.no_grad():
        hip(x)
    hip.eval()
    with torch.no_grad():
        assert torch.equal(hip(y), cold(y))
    # 4. the fused trainer's SGD writes the parameters natively
    tr = flair_amd.SegTrainer(hip.train(), lr=0.05)
    p0 = tr.predict(y)
    tr.train_step(x, lab)
    hip.eval()
    with torch.no_grad():
        assert torch.equal(hip(y), cold(y))
    assert p0.shape == (2, 64, 64)
    # 5. the split entry points on the same eval handle and arena lay the workspace out differently (five re-imported feature
    #    tensors in front of the decoder units): a fused eval forward after them must not take the reuse path
    with torch.no_grad():
        a = hip(y)
        assert torch.equal(hip(y), a)                      # reuse path armed
        feats = hip.encoder(y)
        split = hip.segmentation_head(hip.decoder(*feats))
        again = hip(y)
        assert torch.equal(again, cold(y))
        assert torch.allclose(split, again, rtol=1e-4, atol=1e-4)


ROUND2B_SWITCHES = {"FLAIR_WG_CUS": 256, "FLAIR_HALO_PERSIST": 0, "FLAIR_HALO_PM": 0, "FLAIR_BNR_HALO": 0, "FLAIR_DBIAS_FUSE": 0,
                    "FLAIR_STEM_FUSE": 0, "FLAIR_POOL_BNR": 0}


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_fused_and_persistent_paths_agree_with_the_plain_ones(dev, dtype):
    """The second half of round 2 folded work into neighbouring kernels (stem BN-backward apply inside its weight gradient,
    the stem's BN-backward sums inside the max-pool backward, the head's bias gradient inside its weight gradient, the
    BN-backward reduction inside the small-channel data gradients) and moved the small-channel layers to persistent kernels with a
    register epilogue.  Every such path has a run-time switch back to the plain kernels: one training step each way from the
    same weights must give the same loss and gradients up to summation order (fp32) / bf16 rounding of reordered sums."""
    import flair_amd
    from flair_amd import _lib as L
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 5, 256, 512, generator=g).to(dev)
    lab = torch.randint(0, 13, (2, 256, 512), generator=g).to(torch.uint8).to(dev)
    res = []
    try:
        for plain in (False, True):
            if plain:
                for k, v in ROUND2B_SWITCHES.items():
                    L.check(L.lib().flair_tune_set(k.encode(), v))
            _, m = _pair(5, 13, 11, dev, dtype)   # a fresh handle: its workspace plan belongs to the switch setting
            tr = flair_amd.SegTrainer(m.train(), lr=0.0)
            loss = tr.train_step(x, lab)
            torch.cuda.synchronize()
            res.append((loss.item(), tr.grads.clone(), m.flat_buffers().clone(), m))
    finally:
        defaults = {"FLAIR_WG_CUS": 0, "FLAIR_HALO_PERSIST": 1, "FLAIR_HALO_PM": 1, "FLAIR_BNR_HALO": 1, "FLAIR_DBIAS_FUSE": 1,
                    "FLAIR_STEM_FUSE": 1, "FLAIR_POOL_BNR": 1}
        for k, v in defaults.items():
            L.lib().flair_tune_set(k.encode(), v)
    (l0, g0, b0, m0), (l1, g1, b1, _) = res
    tol = 2e-5 if dtype == "f32" else 2e-2
    assert abs(l0 - l1) <= (1e-6 if dtype == "f32" else 2e-3) * max(1.0, abs(l1)), (l0, l1)
    assert torch.allclose(b0, b1, rtol=1e-5 if dtype == "f32" else 1e-2, atol=1e-6 if dtype == "f32" else 1e-3)   # BN running statistics
    # per parameter tensor: max |difference| relative to the tensor's own scale
    names = [n for n, _ in m0.named_parameters()]
    worst, worst_cos = (0.0, ""), (1.0, "")
    for name, a, b in zip(names, m0._grad_views(g0), m0._grad_views(g1)):
        scale = float(b.abs().max()) + 1e-12
        err = float((a - b).abs().max()) / scale
        if err > worst[0]:
            worst = (err, name)
        cos = float(torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0))
        if cos < worst_cos[0]:
            worst_cos = (cos, name)
    print(f"[{dtype}] worst relative max-difference {worst}, worst cosine {worst_cos}")
    if dtype == "f32":
        assert worst[0] <= tol, worst
    else:
        # bf16: the two settings round different intermediate sums (BatchNorm statistics of the small-channel layers are summed
        # in another order), and every later value inherits that at the 2^-8 level; cancellation-heavy gradients (BatchNorm
        # biases) move by percents of their own scale.  Direction and the bulk must agree.
        assert worst_cos[0] >= 0.995, worst_cos
        assert worst[0] <= 0.10, worst


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 256, 512), (1, 96, 160)])
def test_fused_bn_backward_apply_is_bit_identical_to_the_separate_pass(dev, dtype, shape):
    """Round 3: for the units whose data gradient runs on the halo-GEMM, the kernel that completes the gradient of the unit's
    ReLU output stores it masked (ConvArgs::bnr_mask), the unit's own data gradient applies  dy = k1*dz + k2*y + k3  while it
    stages its halo and writes dy once for the weight-gradient kernel (ConvArgs::ap_*), and the identity branch of a BasicBlock
    reads the masked gradient as the addend of conv1's data gradient (ConvArgs::acc_src): bn_bwd_apply does not run for them.
    The arithmetic and every rounding are those of bn_bwd_apply_kernel, so FLAIR_BWD_FUSE=0 (the separate pass, torch's
    BatchNorm2d backward restated at oracle/unet_resnet34.py) must give the SAME BITS: loss, every gradient, running statistics.
    The second shape is ragged for the tile kernels (fall-backs inside the same step)."""
    import flair_amd
    from flair_amd import _lib as L
    g = torch.Generator().manual_seed(5)
    B, H, W = shape
    x = torch.randn(B, 5, H, W, generator=g).to(dev)
    lab = torch.randint(0, 13, (B, H, W), generator=g).to(torch.uint8).to(dev)
    res = []
    try:
        for fuse in (2, 1, 0):   # 2: apply inside the data gradient too; 1 (default): masked gradients + identity hand-over; 0: round 2
            L.check(L.lib().flair_tune_set(b"FLAIR_BWD_FUSE", fuse))
            _, m = _pair(5, 13, 11, dev, dtype)
            tr = flair_amd.SegTrainer(m.train(), lr=0.0)
            loss = tr.train_step(x, lab)
            torch.cuda.synchronize()
            res.append((loss.item(), tr.grads.clone(), m.flat_buffers().clone()))
    finally:
        L.lib().flair_tune_set(b"FLAIR_BWD_FUSE", 1)
    (l1, g1, b1) = res[-1]
    assert torch.isfinite(g1).all()
    for mode, (l0, g0, b0) in zip((2, 1), res[:2]):
        assert l0 == l1
        assert torch.equal(b0, b1)
        nz = int((g0 != g1).sum())
        assert nz == 0, f"mode {mode}: {nz} of {g0.numel()} gradient elements differ, max |d| {float((g0 - g1).abs().max()):.3e}"


def test_small_eval_forwards_replay_a_hip_graph_with_identical_results(dev):
    """Launch-bound eval forwards (predict_step at the reference's batch size 1, data_module.py:100) are captured into a HIP
    graph on their third identical call and replayed: same bits as the eager call for logits, argmax and probability, fresh
    inputs honoured, and any weight change drops the graph."""
    import flair_amd
    _, hip = _pair(5, 13, 19, dev)
    hip.eval()
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(1, 5, 128, 128, generator=g).to(dev) for _ in range(5)]
    os.environ["FLAIR_EVAL_GRAPH"] = "0"
    try:
        with torch.no_grad():
            want = [(hip(x).clone(), *[t.clone() for t in hip.predict_classes(x, want_prob=True)]) for x in xs]
    finally:
        os.environ.pop("FLAIR_EVAL_GRAPH")
    hip._eval_key = None
    with torch.no_grad():
        for rep in range(2):
            for x, (lg, pr, pb) in zip(xs, want):
                assert torch.equal(hip(x), lg)
                p, q = hip.predict_classes(x, want_prob=True)
                assert torch.equal(p, pr) and torch.equal(q, pb)
        assert len(hip._eval_graphs) == 2                      # logits, and preds + probability
        next(iter(hip.parameters())).mul_(1.25)                # a torch in-place write: version counter moves
        a = hip(xs[0])
        assert len(hip._eval_graphs) == 0 and not torch.equal(a, want[0][0])
        m2 = flair_amd.create_model("unet", "resnet34", encoder_weights=None, in_channels=5, classes=13, compute_dtype="f32")
        m2.load_state_dict(hip.state_dict())
        assert torch.equal(a, m2.to(dev).eval()(xs[0]))


@pytest.mark.parametrize("dtype,classes", [("f32", 13), ("bf16", 13), ("bf16", 19), ("f32", 19)])
def test_predict_argmax_from_the_head_epilogue_equals_argmax_of_the_logits(dev, dtype, classes):
    """SegTrainer.predict takes the argmax out of the head convolution's epilogue (flair_unet_want_preds).  It must equal the
    first-maximum argmax of the logits the same forward would have returned (fp32 NCHW copy of the same rounded values), for
    the one-block (13 classes) and the two-block (19 classes) column layouts."""
    import flair_amd
    _, m = _pair(5, classes, 23, dev, dtype)
    m = m.eval()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 5, 96, 160, generator=g).to(dev)
    tr = flair_amd.SegTrainer(m, lr=0.0)
    preds = tr.predict(x)
    logits = m._c_forward(x, training=False, want_logits=True)
    ref = logits.argmax(dim=1).to(torch.uint8)
    assert preds.shape == ref.shape and preds.dtype == torch.uint8
    assert torch.equal(preds, ref), int((preds != ref).sum())
