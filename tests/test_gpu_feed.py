"""GPU parity of the rows either side of the hot path, through the C ABI (flair_feed_tiles, flair_detect_convert,
flair_confmat_masks) and the host mirrors built on them (TileFeed, metrics, predictionwriter, zone_detect.inference).
Byte and integer work is asserted bit-exact against oracle/data_feed.py and the reference-made goldens; the softmax
under convert() is floating point: argmax exact away from ties (top-2 gap > 1e-6), max-prob within 1e-6, and the
truncated (p * 255) bytes within 1 with >= 99.9 % exact (tolerances stated here, per north_star)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MEANS = [105.08, 110.87, 101.82, 106.38, 53.26]
STDS = [52.17, 45.38, 44, 39.69, 79.3]


def _classes(C, zero=()):
    return {i + 1: [0 if i in zero else 1, f"class{i + 1}"] for i in range(C)}


def test_feed_matches_reference_dataset_bit_exact(dev, golden_dir):
    from flair_amd.data_feed import TileFeed
    g = np.load(os.path.join(golden_dir, "feed_reference.npz"))
    img = torch.from_numpy(g["img_u8"])[None].to(dev)
    raw = torch.from_numpy(g["msk_raw"])[None].to(dev)
    for C in (13, 19):
        b = TileFeed([1, 2, 3, 4, 5], C, "custom", MEANS, STDS)(img, raw)
        assert b["img"].dtype == torch.float32 and b["msk"].dtype == torch.uint8
        assert np.array_equal(b["img"][0].cpu().numpy(), g[f"custom_img_c{C}"])
        assert np.array_equal(b["msk"][0].cpu().numpy(), np.argmax(g[f"msk_onehot_c{C}"], axis=0))
    b = TileFeed([3, 1, 5], 13, "custom", [1.5, 2.25, 100.0], [3.0, 0.7, 51.3])(img)
    assert "msk" not in b and np.array_equal(b["img"][0].cpu().numpy(), g["custom_img_ch315"])
    b = TileFeed([1, 2, 3, 4, 5], 13, "without")(img, ids=["t/IMG_000001.tif"])
    assert sorted(b) == ["id", "img"] and np.array_equal(b["img"][0].cpu().numpy(), g["without_img"])


def test_feed_all_modes_and_d4_against_oracle(dev):
    from flair_amd.data_feed import TileFeed, pack_d4
    from oracle import data_feed as F
    rng = np.random.default_rng(5)
    B, H = 16, 32
    img = rng.integers(0, 256, size=(B, 5, H, H), dtype=np.uint8)
    raw = rng.integers(0, 25, size=(B, H, H), dtype=np.uint8)
    draws = [(v, h, k) for v in (0, 1) for h in (0, 1) for k in range(4)]
    d4 = torch.tensor([pack_d4(*d) for d in draws], dtype=torch.uint8)
    for mode in ("custom", "scaling", "without"):
        feed = TileFeed([1, 2, 3, 4, 5], 19, mode, MEANS, STDS)
        b = feed(torch.from_numpy(img).to(dev), torch.from_numpy(raw).to(dev), d4=d4)
        for s, (v, h, k) in enumerate(draws):
            want_img = F.norm_np(F.d4_np(img[s], v, h, k), mode, MEANS, STDS)
            want_lab = F.d4_np(F.labels_from_raw(raw[s], 19), v, h, k)
            assert np.array_equal(b["img"][s].cpu().numpy(), want_img), (mode, v, h, k)
            assert np.array_equal(b["msk"][s].cpu().numpy(), want_lab), (mode, v, h, k)


def test_feed_full_batch_properties(dev):
    """32 x 5 x 512 x 512 (BASELINE config 2): size-independent properties instead of a CPU comparison."""
    from flair_amd.data_feed import TileFeed, pack_d4
    g = torch.Generator(device="cpu").manual_seed(2022)
    B, S = 32, 512
    img = torch.randint(0, 256, (B, 5, S, S), dtype=torch.uint8, generator=g).to(dev)
    raw = torch.randint(0, 16, (B, S, S), dtype=torch.uint8, generator=g).to(dev)
    plain = TileFeed([1, 2, 3, 4, 5], 13, "without")
    base = plain(img, raw)
    assert torch.equal(base["img"], img.float())
    lab = (raw - 1)  # uint8 wrap
    assert torch.equal(base["msk"], torch.where(lab < 13, lab, torch.zeros_like(lab)))
    # a symmetry followed by its inverse is the identity; sums and label histograms are invariant
    fwd = torch.tensor([pack_d4(s % 2, (s // 2) % 2, s % 4) for s in range(B)], dtype=torch.uint8)
    aug = plain(img, raw, d4=fwd)
    assert torch.equal(aug["img"].long().sum(dim=(2, 3)), base["img"].long().sum(dim=(2, 3)))
    assert torch.equal(torch.bincount(aug["msk"].flatten().long(), minlength=13), torch.bincount(base["msk"].flatten().long(), minlength=13))
    back = plain(aug["img"].to(torch.uint8), None, d4=torch.tensor([pack_d4(0, 0, (4 - s % 4) % 4) for s in range(B)], dtype=torch.uint8))
    back = plain(back["img"].to(torch.uint8), None, d4=torch.tensor([pack_d4(s % 2, (s // 2) % 2, 0) for s in range(B)], dtype=torch.uint8))
    assert torch.equal(back["img"], base["img"])
    # custom normalisation is a per-band function of the byte: check through a 256-entry table made on the host
    feed = TileFeed([1, 2, 3, 4, 5], 13, "custom", MEANS, STDS)
    out = feed(img)["img"]
    for c in range(5):
        table = torch.from_numpy(((np.arange(256, dtype=np.float64) - MEANS[c]) / STDS[c]).astype(np.float32)).to(dev)
        assert torch.equal(out[:, c], table[img[:, c].long()])


def test_feed_errors(dev):
    from flair_amd._lib import FlairHipError
    from flair_amd.data_feed import TileFeed
    feed = TileFeed([1, 2, 6], 13, "scaling")
    with pytest.raises(ValueError):
        feed(torch.zeros(1, 5, 8, 8, dtype=torch.uint8, device=dev))  # band 6 of 5
    feed = TileFeed([1], 13, "scaling")
    with pytest.raises(ValueError):
        feed(torch.zeros(1, 1, 8, 8, dtype=torch.float32, device=dev))
    with pytest.raises(FlairHipError):  # rot90 on a non-square tile
        feed(torch.zeros(1, 1, 8, 16, dtype=torch.uint8, device=dev), d4=torch.tensor([4], dtype=torch.uint8))
    with pytest.raises(FlairHipError):  # W % 4
        feed(torch.zeros(1, 1, 6, 6, dtype=torch.uint8, device=dev))


def _check_convert(got_am, got_cp, logits, margin, oracle):
    want_am = oracle.detect_tile_np(logits, margin, "argmax")
    want_cp = oracle.detect_tile_np(logits, margin, "class_prob")
    x = torch.softmax(torch.from_numpy(logits).double(), 0).numpy()
    S = x.shape[-1]
    x = x[:, margin:S - margin, margin:S - margin]
    top2 = np.sort(x, axis=0)[-2:]
    clear = (top2[1] - top2[0]) > 1e-6
    assert np.array_equal(got_am[0][clear], want_am[0][clear])
    assert (got_am[0] != want_am[0]).mean() < 1e-4
    assert np.abs(got_am[1] - want_am[1]).max() < 1e-6
    diff = np.abs(got_cp.astype(int) - want_cp.astype(int))
    assert diff.max() <= 1 and (diff == 0).mean() >= 0.999


def test_detect_convert_matches_reference_convert(dev, golden_dir):
    from flair_amd.zone_detect import detect_convert
    from oracle import data_feed as F
    d = np.load(os.path.join(golden_dir, "detect_convert_reference.npz"))
    lg = torch.from_numpy(d["logits"])[None].to(dev)
    am = detect_convert(lg, 0, "argmax")[0].cpu().numpy()
    cp = detect_convert(lg, 0, "class_prob")[0].cpu().numpy()
    assert am.dtype == np.float32 and am.shape == (2, 40, 40) and cp.dtype == np.uint8 and cp.shape == (19, 40, 40)
    # against the reference function's own outputs (torch softmax -> zone_detect convert)
    assert np.array_equal(am[0], d["argmax"][0]) and np.abs(am[1] - d["argmax"][1]).max() < 1e-6
    diff = np.abs(cp.astype(int) - d["class_prob"].astype(int))
    assert diff.max() <= 1 and (diff == 0).mean() >= 0.999
    _check_convert(am, cp, d["logits"], 0, F)
    # margin crop (compare.py:71-75) and batches
    rng = np.random.default_rng(1)
    logits = (rng.standard_normal((3, 13, 64, 64)) * 4).astype(np.float32)
    am = detect_convert(torch.from_numpy(logits).to(dev), 16, "argmax").cpu().numpy()
    cp = detect_convert(torch.from_numpy(logits).to(dev), 16, "class_prob").cpu().numpy()
    assert am.shape == (3, 2, 32, 32) and cp.shape == (3, 13, 32, 32)
    for b in range(3):
        _check_convert(am[b], cp[b], logits[b], 16, F)
    # exact ties keep the first maximum, like np.argmax
    tie = torch.zeros(1, 4, 4, 4, device=dev)
    tie[0, 1] = tie[0, 3] = 2.0
    assert (detect_convert(tie, 0, "argmax")[0, 0] == 1).all()
    with pytest.raises(ValueError):
        detect_convert(tie, 2, "argmax")
    with pytest.raises(ValueError):
        detect_convert(tie, 0, "probabilities")


def test_confmat_masks_vs_oracle(dev, golden_dir):
    from flair_amd.metrics import MaskConfusion
    from oracle import seg_step
    rng = np.random.default_rng(9)
    for C, n in ((13, 512 * 512), (19, 100003), (13, 7), (19, 16)):
        truth = rng.integers(0, C + 4, size=n, dtype=np.uint8)  # stored 0 and > C occur
        pred = rng.integers(0, C + 2, size=n, dtype=np.uint8)
        want = seg_step.confusion_matrix_np(truth.astype(np.int64) - 1, pred, C)
        acc = MaskConfusion(C, device=dev)
        acc.update(torch.from_numpy(truth).to(dev), torch.from_numpy(pred).to(dev))
        assert np.array_equal(acc.compute(), want), (C, n)
        acc.update(torch.from_numpy(truth).to(dev), torch.from_numpy(pred).to(dev))
        assert np.array_equal(acc.compute(), 2 * want)
    # the reference recipe behind mIoU 56.54 (metrics_reference.json), stored labels = class + 1
    g = json.load(open(os.path.join(golden_dir, "metrics_reference.json")))["c13_seed0"]
    rng = np.random.default_rng(0)
    t = rng.integers(0, 13, size=512 * 512)
    p = np.where(rng.random(t.size) < 0.7, t, rng.integers(0, 13, size=t.size))
    acc = MaskConfusion(13, device=dev)
    acc.update(torch.from_numpy((t + 1).astype(np.uint8)).to(dev), torch.from_numpy(p.astype(np.uint8)).to(dev))
    assert acc.compute().tolist() == g["confmat"]
    # full size: 32 tiles of 512 x 512 in one launch; every in-range pair is counted exactly once
    T = torch.randint(0, 16, (32, 512, 512), dtype=torch.uint8, device=dev)
    P = torch.randint(0, 14, (32, 512, 512), dtype=torch.uint8, device=dev)
    acc = MaskConfusion(13, device=dev)
    acc.update(T, P)
    cm = acc.confmat
    valid = ((T - 1) < 13) & (P < 13)
    assert int(cm.sum()) == int(valid.sum())
    assert torch.equal(cm.sum(1), torch.bincount((T - 1)[valid].long(), minlength=13))


def test_metrics_and_writer_end_to_end(dev, tmp_path):
    """predict -> PRED_*.tif (writer.py) -> metrics.json / confmat.npy (metrics.py) on synthetic tiles."""
    from PIL import Image
    from flair_amd import metrics as M
    from flair_amd.writer import predictionwriter
    from oracle import seg_step
    C, n, S = 19, 6, 64
    classes = _classes(C, zero=(14, 15, 16, 18))
    rng = np.random.default_rng(3)
    truth = rng.integers(0, C + 2, size=(n, S, S), dtype=np.uint8)
    preds = rng.integers(0, C, size=(n, S, S))
    (tmp_path / "gt").mkdir()
    rows = []
    for i in range(n):
        Image.fromarray(truth[i]).save(tmp_path / "gt" / f"MSK_{i:06d}.tif")
        rows.append(f"/data/x/img/IMG_{i:06d}.tif,{tmp_path}/gt/MSK_{i:06d}.tif")
    rows.append(f"/data/x/img/IMG_999999.tif,{tmp_path}/gt/MSK_missing.tif")  # reported and skipped, like the reference
    (tmp_path / "test.csv").write_text("\n".join(rows) + "\n")
    out_dir = tmp_path / "run" / "predictions"
    w = predictionwriter({"georeferencing_output": False}, str(out_dir), "batch")
    for i in range(0, n, 2):
        batch = {"preds": torch.from_numpy(preds[i:i + 2]).to(dev), "id": [f"/data/x/img/IMG_{j:06d}.tif" for j in (i, i + 1)]}
        w.on_predict_batch_end(None, None, batch, batch, i // 2)
    # only the hook sequence Lightning drives (tasks.py:125-142 builds the callback locally, main.py:238-242 reads the files
    # right after trainer.predict): no explicit flush()/close() by the caller
    w.on_predict_epoch_end(None, None)
    w.on_predict_end(None, None)
    assert w._worker is None
    for i in range(n):
        assert np.array_equal(np.asarray(Image.open(out_dir / f"PRED_IMG_{i:06d}.tif")), preds[i].astype(np.uint8))
    with pytest.raises(RuntimeError):
        predictionwriter({"georeferencing_output": True}, str(out_dir), "batch").write_on_batch_end(None, None, batch, None, batch, 0, 0)
    config = {"paths": {"test_csv": str(tmp_path / "test.csv")}, "classes": classes}
    out = M.metrics(config, out_dir, device=dev)
    cm = seg_step.confusion_matrix_np(truth.astype(np.int64) - 1, preds, C)
    assert np.array_equal(np.load(tmp_path / "run" / "metrics" / "confmat.npy"), cm)
    want = seg_step.offline_metrics(cm, [classes[k][0] for k in classes])
    saved = json.load(open(tmp_path / "run" / "metrics" / "metrics.json"))
    assert list(saved) == ["Avg_metrics_name", "Avg_metrics", "classes", "per_class_iou", "per_class_fscore",
                           "per_class_precision", "per_class_recall"]  # metrics.py:90-108
    assert saved["Avg_metrics"] == [float(v) for v in want["Avg_metrics"]] == [float(v) for v in out["Avg_metrics"]]
    assert saved["per_class_iou"] == [float(v) for v in want["per_class_iou"]] and len(saved["classes"]) == 15


def test_writer_without_close_neither_hangs_nor_loses_tiles(tmp_path):
    """the worker thread must not keep the interpreter alive (a non-daemon queue worker is joined before the
    atexit hooks run) and a tile queued by a bare write_on_batch_end — no close(), no on_predict_end — must still be on disk
    after the process has exited."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys, torch
sys.path[:0] = [r'{root}', r'{os.path.join(root, "flair-1_amd")}']
from flair_amd.writer import predictionwriter
w = predictionwriter({{"georeferencing_output": False}}, r'{tmp_path}', "batch")
p = (torch.arange(2 * 32 * 32, device="cuda:0") % 19).reshape(2, 32, 32)
w.write_on_batch_end(None, None, {{"preds": p, "id": ["a/IMG_000001.tif", "a/IMG_000002.tif"]}}, None, None, 0, 0)
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr[-2000:]
    from PIL import Image
    want = (np.arange(2 * 32 * 32) % 19).reshape(2, 32, 32).astype(np.uint8)
    for i in (1, 2):
        assert np.array_equal(np.asarray(Image.open(tmp_path / f"PRED_IMG_00000{i}.tif")), want[i - 1])


def test_zone_detect_inference_fused(dev):
    import flair_amd
    from flair_amd.zone_detect import inference
    from oracle import data_feed as F
    from oracle import unet_resnet34 as om
    ref = om.seeded_model(5, 19, 4).eval()
    hip = flair_amd.create_model("unet", "resnet34", encoder_weights=None, in_channels=5, classes=19, compute_dtype="f32")
    hip.load_state_dict(ref.state_dict(), strict=True)
    hip = hip.to(dev).eval()
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 5, 64, 64, generator=g)
    samples = {"image": x, "index": torch.tensor([3, 11])}
    cfg = {"margin": 8, "output_type": "argmax", "model_framework": {"model_provider": "SegmentationModelsPytorch"}}
    pred, idx = inference(dev, hip, True, cfg, samples)
    assert idx.tolist() == [3, 11] and pred.shape == (2, 2, 48, 48) and pred.dtype == np.float32
    with torch.no_grad():
        lg = ref(x).numpy()
    from oracle import parity
    for b in range(2):
        want = F.detect_tile_np(lg[b], 8, "argmax")
        parity.assert_mask_parity(f"zone_detect_inference_tile{b}", want[0], pred[b, 0], parity.top2_gap(lg[b][:, 8:-8, 8:-8], axis=0))
        assert np.abs(pred[b, 1] - want[1]).max() < 2e-3
    cfg["output_type"] = "class_prob"
    pred, _ = inference(dev, hip, True, cfg, samples)
    assert pred.shape == (2, 19, 48, 48) and pred.dtype == np.uint8
    want = F.detect_tile_np(lg[0], 8, "class_prob")
    assert np.abs(pred[0].astype(int) - want.astype(int)).max() <= 1
    full, _ = inference(dev, hip, True, cfg, samples, fused=False)  # the reference's return value: all probabilities
    assert full.shape == (2, 19, 64, 64) and abs(full.sum(1).mean() - 1) < 1e-5
    with pytest.raises(RuntimeError):
        inference(torch.device("cpu"), hip, False, cfg, samples)


def test_config1_plumbing_on_synthetic_tiles(dev, tmp_path):
    """SURVEY.md §8d config 1 on synthetic tiles: feed -> fit (train/val hooks) -> predict -> writer -> metrics.json."""
    from PIL import Image
    from flair_amd import metrics as M
    from flair_amd.tasks_utils import get_segmentation_module, get_tile_feed
    from flair_amd.writer import predictionwriter
    S, C = 64, 13
    classes = _classes(C)
    config = {"model_framework": {"model_provider": "SegmentationModelsPytorch",
                                  "SegmentationModelsPytorch": {"encoder_decoder": "resnet34_unet", "encoder_weights": None}},
              "use_metadata": False, "use_augmentation": True, "use_weights": True, "channels": [1, 2, 3], "classes": classes,
              "norm_type": "scaling", "norm_means": [], "norm_stds": [], "learning_rate": 0.02, "batch_size": 2,
              "georeferencing_output": False}
    g = torch.Generator().manual_seed(2022)
    def tiles(n):
        img = torch.randint(0, 256, (n, 5, S, S), dtype=torch.uint8, generator=g)
        raw = torch.randint(1, C + 1, (n, S, S), dtype=torch.uint8, generator=g)
        raw[:, : S // 2] = (img[:, 0, : S // 2] // 32 + 1)  # a learnable relation between band 1 and the label
        return img, raw
    train, val, test = tiles(8), tiles(4), tiles(4)
    torch.manual_seed(2022)
    module = get_segmentation_module(config, "train", compute_dtype="f32").to(dev)
    feed_train = get_tile_feed(config, train=True, generator=torch.Generator().manual_seed(1))
    feed_eval = get_tile_feed(config, train=False)
    module.setup("fit")
    opt = module.configure_optimizers()["optimizer"]
    epoch_losses = []
    for epoch in range(6):
        module.train()
        for i in range(0, 8, 2):
            batch = feed_train(train[0][i:i + 2].to(dev), train[1][i:i + 2].to(dev))
            opt.zero_grad()
            loss = module.training_step(batch, i // 2)
            loss.backward()
            opt.step()
        module.on_train_epoch_end()
        epoch_losses.append(float(module.train_epoch_loss))
        module.eval()
        with torch.no_grad():
            for i in range(0, 4, 2):
                module.validation_step(feed_eval(val[0][i:i + 2].to(dev), val[1][i:i + 2].to(dev)), i // 2)
        module.on_validation_epoch_end()
        assert 0.0 <= float(module.val_epoch_metrics) <= 1.0
    assert all(np.isfinite(epoch_losses)) and epoch_losses[-1] < epoch_losses[0]
    # predict + write + offline metrics
    pmod = get_segmentation_module(config, "predict", compute_dtype="f32")
    pmod.model.load_state_dict(module.model.state_dict())
    pmod = pmod.to(dev).eval()
    out_dir = tmp_path / "out" / "predictions"
    w = predictionwriter(config, str(out_dir), "batch")
    (tmp_path / "gt").mkdir()
    rows = []
    for i in range(4):
        Image.fromarray(test[1][i].numpy()).save(tmp_path / "gt" / f"MSK_{i}.tif")
        rows.append(f"/d/IMG_{i}.tif,{tmp_path}/gt/MSK_{i}.tif")
    (tmp_path / "test.csv").write_text("\n".join(rows) + "\n")
    with torch.no_grad():
        for i in range(4):  # predict batch size 1 (data_module.py:100)
            batch = feed_eval(test[0][i:i + 1].to(dev), ids=[f"/d/IMG_{i}.tif"])
            res = pmod.predict_step(batch, i)
            assert res["preds"].shape == (1, S, S) and res["preds"].dtype == torch.int64
            w.on_predict_batch_end(None, pmod, res, batch, i)
    w.close()
    config["paths"] = {"test_csv": str(tmp_path / "test.csv")}
    out = M.metrics(config, out_dir, device=dev)
    saved = json.load(open(tmp_path / "out" / "metrics" / "metrics.json"))
    assert list(saved) == list(out) and len(saved["per_class_iou"]) == C
    assert 0 <= saved["Avg_metrics"][0] <= 100


def test_zone_detector_whole_raster_vs_oracle(dev):
    """Default zone_detect pipeline on a raster that is not a multiple of the stride (edge windows pulled back,
    boundless reads outside the raster): device gather -> HIP U-Net -> fused convert + stitch vs the sequential CPU
    restatement (oracle/zone_detect.py)."""
    import flair_amd
    from flair_amd.zone_detect import ZoneDetector
    from oracle import unet_resnet34 as om
    from oracle import zone_detect as oz
    C = 19
    ref = om.seeded_model(5, C, 6).eval()
    hip = flair_amd.create_model("unet", "resnet34", encoder_weights=None, in_channels=5, classes=C, compute_dtype="f32")
    hip.load_state_dict(ref.state_dict(), strict=True)
    hip = hip.to(dev).eval()
    rng = np.random.default_rng(12)
    raster = rng.integers(0, 256, size=(5, 168, 200), dtype=np.uint8)
    cfg = {"img_pixels_detection": 64, "margin": 8, "output_type": "argmax", "n_classes": C, "batch_size": 4,
           "channels": [1, 2, 3, 4, 5], "norma_task": [{"norm_type": "custom", "norm_means": MEANS, "norm_stds": STDS}]}
    from oracle import parity
    gap = np.zeros((168, 200))
    want = oz.detect_raster_np(ref, raster, cfg, gap_out=gap)
    got = ZoneDetector(hip, cfg).run(torch.from_numpy(raster).to(dev)).cpu().numpy()
    assert got.shape == want.shape == (2, 168, 200) and got.dtype == np.float32
    parity.assert_mask_parity("zone_detector_raster_168x200", want[0], got[0], gap)
    assert np.abs(got[1] - want[1]).max() < 2e-3
    cfg2 = dict(cfg, output_type="class_prob", channels=[3, 1, 5, 2, 4], batch_size=7)
    want = oz.detect_raster_np(ref, raster, cfg2)
    got = ZoneDetector(hip, cfg2).run(torch.from_numpy(raster).to(dev)).cpu().numpy()
    assert got.shape == (C, 168, 200) and got.dtype == np.uint8
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
    with pytest.raises(NotImplementedError):
        ZoneDetector(hip, dict(cfg, overlap_strat=True, strategies={"tiling": {"stride_range": [0.5]}}))
    with pytest.raises(RuntimeError):
        ZoneDetector(hip, cfg).run(torch.from_numpy(raster))


def test_gather_tiles_boundless_and_normalised(dev):
    """The window read alone, bit-exact: zero fill outside the raster happens BEFORE normalisation."""
    import ctypes as C
    from flair_amd import _lib as L
    from oracle import data_feed as F
    from oracle import zone_detect as oz
    rng = np.random.default_rng(2)
    raster = rng.integers(0, 256, size=(5, 100, 90), dtype=np.uint8)
    S = 32
    origins = [(-8, -8), (70, 80), (0, 0), (-31, 99), (58, 68)]
    tiles = torch.tensor([[x, y, 0, 0, 0, 0] for x, y in origins], dtype=torch.int32, device=dev)
    rd = torch.from_numpy(raster).to(dev)
    ch = [2, 5, 1]
    out = torch.empty(len(origins), 3, S, S, device=dev)
    L.check(L.lib().flair_gather_tiles(L.ptr(rd), 5, 100, 90, L.ptr(tiles), len(origins), S, (C.c_int * 3)(*ch), 3, 2,
                                       (C.c_double * 3)(*MEANS[:3]), (C.c_double * 3)(*STDS[:3]), L.ptr(out), L.stream()))
    for b, (x, y) in enumerate(origins):
        win = oz.read_window_np(raster, ch, float(x), float(100 - y - S), S)
        assert np.array_equal(out[b].cpu().numpy(), F.norm_np(win, "custom", MEANS[:3], STDS[:3])), (x, y)
