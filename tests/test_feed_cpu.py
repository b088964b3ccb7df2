"""CPU suite for the rows either side of the hot path: the oracle restatement
(oracle/data_feed.py) against goldens produced by the reference's own data_loader.norm / read_msk /
parsing_metadata / zone_detect convert (tests/golden/make_golden_feed.py), and the host-side mirrors that need no
GPU (metadata encoding, the metrics.json family, D4 draws)."""
import json
import os
import tempfile

import numpy as np
import pytest
import torch

from oracle import data_feed as F
from oracle import seg_step


@pytest.fixture(scope="module")
def feed(golden_dir):
    return np.load(os.path.join(golden_dir, "feed_reference.npz"))


def test_oracle_norm_and_labels_match_reference_dataset(feed):
    img, raw = feed["img_u8"], feed["msk_raw"]
    assert set(np.unique(img)) == set(range(256))  # every byte value exercised
    for C in (13, 19):
        assert np.array_equal(F.norm_np(img, "custom", feed["means"], feed["stds"]), feed[f"custom_img_c{C}"])
        onehot = F.onehot_from_raw(raw, C)
        assert np.array_equal(onehot.astype(np.uint8), feed[f"msk_onehot_c{C}"])
        # all-zero one-hot pixels (raw 0 or > C) decode to class 0
        lab = F.labels_from_raw(raw, C)
        assert np.array_equal(lab, np.argmax(feed[f"msk_onehot_c{C}"], axis=0))
        assert (lab[(raw == 0) | (raw > C)] == 0).all() and ((raw == 0) | (raw > C)).any()
    assert np.array_equal(F.norm_np(img[[2, 0, 4]], "custom", [1.5, 2.25, 100.0], [3.0, 0.7, 51.3]), feed["custom_img_ch315"])
    assert np.array_equal(F.norm_np(img, "without"), feed["without_img"])
    with pytest.raises(SystemExit):
        F.norm_np(img, "minmax")


def test_oracle_scaling_is_unpinned_but_sane(feed):
    s = F.norm_np(feed["img_u8"], "scaling")
    assert s.dtype == np.float32 and s.min() == 0.0 and s.max() == 1.0
    assert np.abs(s - feed["img_u8"].astype(np.float32) / 255).max() <= 6e-8


def test_d4_is_the_dihedral_group():
    a = np.arange(2 * 6 * 6).reshape(2, 6, 6)
    seen = {F.d4_np(a, v, h, k).tobytes() for v in (0, 1) for h in (0, 1) for k in range(4)}
    assert len(seen) == 8  # 16 draws, 8 distinct symmetries
    assert np.array_equal(F.d4_np(a, 0, 0, 1), np.rot90(a, 1, axes=(1, 2)))
    assert np.array_equal(F.d4_np(a, 1, 0, 0), a[:, ::-1])
    assert np.array_equal(F.d4_np(a, 0, 1, 0), a[:, :, ::-1])
    assert np.array_equal(F.d4_np(a, 1, 1, 0), F.d4_np(a, 0, 0, 2))


def test_metadata_encoding_oracle_and_mirror_match_reference(golden_dir):
    m = json.load(open(os.path.join(golden_dir, "metadata_encoding_reference.json")))
    gold = np.array(m["encoding"])
    assert gold.shape == (4, 45)
    assert np.abs(np.array(F.parsing_metadata_np(m["paths"], m["records"])) - gold).max() <= 1e-15
    from flair_amd import tasks_utils
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "aerial.json")
        json.dump(m["records"], open(p, "w"))
        enc = np.array(tasks_utils.parsing_metadata(m["paths"], {"paths": {"path_metadata_aerial": p}}))
    assert np.abs(enc - gold).max() <= 1e-15
    bad = dict(m["records"]["IMG_061946"], date="2017-01-01")
    with pytest.raises(NameError):
        tasks_utils.encode_metadata_record(bad)


def test_detect_convert_oracle_matches_reference(golden_dir):
    d = np.load(os.path.join(golden_dir, "detect_convert_reference.npz"))
    assert np.array_equal(F.detect_convert_np(d["probs"], "argmax"), d["argmax"])
    assert np.array_equal(F.detect_convert_np(d["probs"], "class_prob"), d["class_prob"])
    assert np.array_equal(F.detect_convert_np(d["tie"], "argmax"), d["argmax_tie"])  # first maximum wins
    full = F.detect_tile_np(d["logits"], 0, "argmax")
    assert np.array_equal(full[0], d["argmax"][0]) and np.abs(full[1] - d["argmax"][1]).max() < 1e-6
    crop = F.detect_tile_np(d["logits"], 8, "class_prob")
    assert crop.shape == (19, 24, 24)
    assert np.abs(crop.astype(int) - d["class_prob"][:, 8:-8, 8:-8]).max() <= 1


def test_metrics_mirror_matches_reference_numbers(golden_dir):
    from flair_amd import metrics as M
    g = json.load(open(os.path.join(golden_dir, "metrics_reference.json")))
    c = g["c13_seed0"]
    cm = np.array(c["confmat"])
    iou, miou = M.class_IoU(cm, 13)
    assert miou == c["miou"] and iou.tolist() == c["iou"]
    assert M.overall_accuracy(cm) == c["oa"]
    pr, mpr = M.class_precision(cm)
    rc, mrc = M.class_recall(cm)
    fs, mfs = M.class_fscore(pr, rc)
    assert (pr.tolist(), rc.tolist(), fs.tolist()) == (c["precision"], c["recall"], c["fscore"])
    assert (mpr, mrc, mfs) == (c["mprecision"], c["mrecall"], c["mfscore"])
    # 19 classes with four weight-0 classes and absent classes (0/0 -> 0)
    z = g["c19_seed1_zero_weight"]
    classes = {i + 1: [w, f"class{i + 1}"] for i, w in enumerate(z["weights"])}
    out = M.metrics_from_confmat(np.array(z["confmat"]), classes)
    assert list(out) == ["Avg_metrics_name", "Avg_metrics", "classes", "per_class_iou", "per_class_fscore",
                         "per_class_precision", "per_class_recall"]
    assert out["Avg_metrics_name"] == ["mIoU", "Overall Accuracy", "Fscore", "Precision", "Recall"]
    assert out["Avg_metrics"] == [z["miou"], z["oa"], z["mfscore"], z["mprecision"], z["mrecall"]]
    assert out["per_class_iou"] == z["iou"] and len(out["classes"]) == 15
    ref = seg_step.offline_metrics(np.array(z["confmat"]), z["weights"])
    assert out["per_class_recall"] == list(ref["per_class_recall"])
    json.dumps(out)  # serialisable as the reference's metrics.json


def test_d4_draws_and_feed_refuses_host_tensors():
    from flair_amd import data_feed as DF
    g = torch.Generator().manual_seed(0)
    f = DF.draw_d4(20000, g)
    assert f.dtype == torch.uint8 and int(f.max()) < 16
    v, h, k = (f & 1).float().mean(), ((f >> 1) & 1).float().mean(), (f >> 2)
    assert abs(v - 0.5) < 0.02 and abs(h - 0.5) < 0.02
    # RandomRotate90(p=.5) with a uniform factor: P(k != 0) = 0.5 * 3/4
    assert abs((k != 0).float().mean() - 0.375) < 0.02
    assert DF.pack_d4(1, 0, 3) == 0b1101
    feed = DF.TileFeed(channels=[1, 2, 3], num_classes=13, norm_type="scaling")
    with pytest.raises(RuntimeError):
        feed(torch.zeros(1, 3, 8, 8, dtype=torch.uint8))
    with pytest.raises(SystemExit):
        DF.TileFeed(norm_type="minmax")
    with pytest.raises(SystemExit):
        DF.TileFeed(channels=[1, 2], norm_type="custom", means=[1.0], stds=[1.0, 2.0])


@pytest.mark.parametrize("W,H,P,m", [(200, 168, 64, 8), (512, 512, 512, 128), (1000, 700, 512, 128), (48, 48, 64, 8), (300, 301, 64, 0),
                                     (2048, 1536, 512, 128)])
def test_zone_detect_tile_grid_owns_each_pixel_once_like_sequential_writes(W, H, P, m):
    """The device stitch writes every window in one launch; its ownership rectangles must reproduce the result of the
    reference's sequential window writes (later windows overwrite earlier ones, main.py:404-421)."""
    from flair_amd.zone_detect import get_stride, tile_grid
    from oracle.zone_detect import slice_extent_np
    assert get_stride({"img_pixels_detection": P, "margin": m}) == [P - 2 * m]
    assert get_stride({"img_pixels_detection": 512, "margin": 128, "overlap_strat": True,
                       "strategies": {"tiling": {"stride_range": [0.25, 0.5]}}}) == [128, 256]
    grid, rows = tile_grid((W, H), P, m), slice_extent_np(W, H, P, m)
    assert len(grid) == len(rows)
    last = np.full((H, W), -1)
    mine = np.full((H, W), -1)
    count = np.zeros((H, W), int)
    for t, (g, r) in enumerate(zip(grid, rows)):
        assert g[0] == round(r["x"]) and g[1] == H - round(r["y"]) - P
        last[H - round(r["top"]):H - round(r["bottom"]), round(r["left"]):round(r["right"])] = t
        mine[g[4]:g[5], g[2]:g[3]] = t
        count[g[4]:g[5], g[2]:g[3]] += 1
    assert (count == 1).all() and (mine == last).all()
    with pytest.raises(ValueError):
        tile_grid((P - 2 * m - 1, H), P, m)


def test_tiling_helpers_match_reference(golden_dir):
    """get_stride / out_of_bounds / get_tile_coord / patch_overlap / patch_weights / total_weights / slice_pixels against
    the reference functions themselves (imported directly by tests/golden/make_golden_feed.py)."""
    from flair_amd import zone_detect as Z
    g = json.load(open(os.path.join(golden_dir, "tiling_reference.json")))
    assert Z.get_stride({"img_pixels_detection": 512, "margin": 128}) == g["get_stride_default"]
    assert Z.get_stride({"img_pixels_detection": 512, "margin": 128, "overlap_strat": True,
                         "strategies": {"tiling": {"stride_range": [0.25, 0.5, 0.75]}}}) == g["get_stride_overlap"]
    for big, box, want in g["out_of_bounds"]:
        assert Z.out_of_bounds(big, box) == want
    assert np.array_equal(Z.patch_weights(7, 0.5, "exp"), np.array(g["patch_weights_exp_7"]))
    assert np.array_equal(Z.patch_weights(8, 0.5, "gaussian"), np.array(g["patch_weights_gauss_8"]))
    for c in g["cases"]:
        size, P, m, st, q = tuple(c["img_size"]), c["patch_size"], c["margin"], c["stride"], c["query"]
        assert Z.get_tile_coord(q[0], q[1], size[0], P, st) == c["tile_coord_x"]
        po = Z.patch_overlap(size, P, q, st)
        assert po.dtype == np.uint8 and np.array_equal(po, np.array(c["patch_overlap"]))
        tw, steps = Z.total_weights(size, P, q, st)
        assert tw.dtype == np.float32 and steps == []
        assert np.abs(tw.astype(np.float64) - np.array(c["total_weights"])).max() < 1e-6   # float32 sums, order of tiles
        assert [list(t) for t in Z.slice_pixels(size, P, m, st)] == c["slice_pixels"]
    toy = g["cases"][0]   # the reference's own 5x5 / patch 3 / stride 2 example: centre pixel seen by all four tiles
    assert np.array(toy["patch_overlap"])[2, 2] == 4
