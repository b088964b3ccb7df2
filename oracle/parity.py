"""TEST INFRASTRUCTURE — the ONE mask-parity rule, used by every comparison of a HIP argmax mask with the oracle's
(tests/, __graft_entry__.smoke()).

north_star: "argmax masks bit-exact vs the CPU reference".  Both sides compute fp32 logits with different summation
orders (MKL-DNN blocks vs the MFMA k-ordered FMA chain), so a pixel whose two best classes are closer than the logits'
own rounding noise has no defined winner.  The rule (task_module.py:71-79 takes argmax(softmax(logits))):
    * ZERO mismatches among pixels whose oracle top-2 softmax-probability gap exceeds GAP = 1e-5;
    * every mismatch is counted and the largest gap among the mismatching pixels is reported;
    * the confusion matrix is compared exactly over the decided pixels.
Each call appends a record to the parity log (FLAIR_PARITY_JSON, default gpurun_out/r3_parity.json under the repo root
when that directory exists), which is copied to profiles/ each round.
"""
import json
import os

import numpy as np

GAP = 1e-5


def top2_gap(logits: np.ndarray, axis: int = 1) -> np.ndarray:
    """Oracle top-2 softmax probability gap per pixel (float64 softmax of the fp32 logits)."""
    x = np.asarray(logits, dtype=np.float64)
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    p = e / e.sum(axis=axis, keepdims=True)
    ps = np.sort(p, axis=axis)
    hi = np.take(ps, -1, axis=axis)
    lo = np.take(ps, -2, axis=axis)
    return hi - lo


def _log_path():
    p = os.environ.get("FLAIR_PARITY_JSON")
    if p:
        return p
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "gpurun_out")
    return os.path.join(d, "r3_parity.json") if os.path.isdir(d) else None


def record(entry: dict):
    path = _log_path()
    if not path:
        return
    try:
        data = json.load(open(path)) if os.path.exists(path) else {}
    except Exception:  # noqa: BLE001
        data = {}
    data[entry["test"]] = entry
    with open(path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)


def assert_mask_parity(name, mask_ref, mask_hip, gap, logits_ref=None, logits_hip=None, labels=None, confmat_hip=None,
                       num_classes=None, gap_thr=GAP):
    """mask_ref / mask_hip: integer arrays of equal shape; gap: the oracle's top-2 probability gap per pixel (same shape).
    Optional: max |logit difference| when both logit tensors are given; exact confusion-matrix check over the decided
    pixels when labels (same shape) are given (confmat_hip: the HIP confusion matrix over ALL pixels — the tie pixels'
    contribution is removed from both sides before comparing)."""
    mask_ref = np.asarray(mask_ref).astype(np.int64)
    mask_hip = np.asarray(mask_hip).astype(np.int64)
    gap = np.asarray(gap)
    assert mask_ref.shape == mask_hip.shape == gap.shape, (mask_ref.shape, mask_hip.shape, gap.shape)
    diff = mask_ref != mask_hip
    flips = int(diff.sum())
    decided = gap > gap_thr
    bad = int((diff & decided).sum())
    entry = {"test": name, "pixels": int(mask_ref.size), "flips": flips, "flips_where_gap_gt_1e-5": bad,
             "max_gap_among_flipped": float(gap[diff].max()) if flips else 0.0,
             "undecided_pixels": int((~decided).sum()), "gap_threshold": gap_thr}
    if logits_ref is not None and logits_hip is not None:
        entry["max_abs_dlogit"] = float(np.abs(np.asarray(logits_ref, np.float64) - np.asarray(logits_hip, np.float64)).max())
    if labels is not None and num_classes is not None:
        lab = np.asarray(labels).astype(np.int64)
        cm_ref = np.bincount((lab[decided] * num_classes + mask_ref[decided]).ravel(), minlength=num_classes ** 2)
        cm_hip = np.bincount((lab[decided] * num_classes + mask_hip[decided]).ravel(), minlength=num_classes ** 2)
        entry["confmat_equal_on_decided"] = bool(np.array_equal(cm_ref, cm_hip))
        if confmat_hip is not None:   # the device-side matrix: remove what the undecided pixels contributed to it
            und = ~decided
            tie = np.bincount((lab[und] * num_classes + mask_hip[und]).ravel(), minlength=num_classes ** 2)
            dev = np.asarray(confmat_hip).astype(np.int64).ravel() - tie
            entry["device_confmat_equal_on_decided"] = bool(np.array_equal(dev, cm_ref))
    record(entry)
    assert bad == 0, entry
    assert entry.get("confmat_equal_on_decided", True) and entry.get("device_confmat_equal_on_decided", True), entry
    return entry


def assert_masks_within_logit_error(name, logits_ref, logits_hip, mask_hip, max_rel_dlogit=5e-3, axis=1, max_rel_rms=None):
    """For comparisons where the two sides do NOT hold identical weights (after a training step each side has applied its
    own fp32 gradient; the two agree to ~1e-4 per weight, and an eval-mode forward of a freshly initialised network —
    running statistics one step old — amplifies that to ~1e-3 of the logit scale): with delta = max |logit difference|,
    the argmax can only differ where the oracle's top-2 LOGIT gap is at most 2 * delta.  Asserts that bound for every
    mismatching pixel (the tight part) and delta <= max_rel_dlogit * max(1, max |logit|) (a sanity bound on the step
    itself, whose parity is asserted elsewhere: loss, |dW|, elementwise gradients); records flips, delta and the largest
    logit gap among the flipped pixels."""
    lr = np.asarray(logits_ref, np.float64)
    lh = np.asarray(logits_hip, np.float64)
    delta = float(np.abs(lr - lh).max())
    srt = np.sort(lr, axis=axis)
    lgap = np.take(srt, -1, axis=axis) - np.take(srt, -2, axis=axis)
    mask_ref = lr.argmax(axis=axis)
    mask_hip = np.asarray(mask_hip).astype(np.int64)
    diff = mask_ref != mask_hip
    flips = int(diff.sum())
    worst = float(lgap[diff].max()) if flips else 0.0
    err = np.abs(lr - lh)
    entry = {"test": name, "pixels": int(mask_ref.size), "flips": flips, "max_abs_dlogit": delta,
             "rms_dlogit": float(np.sqrt((err ** 2).mean())), "p999_abs_dlogit": float(np.quantile(err, 0.999)),
             "max_logit_gap_among_flipped": worst, "logit_scale": float(np.abs(lr).max()),
             "rule": "flip only where top-2 logit gap <= 2 * max|dlogit|; max|dlogit| <= %g * max(1, max|logit|)" % max_rel_dlogit}
    record(entry)
    assert delta <= max_rel_dlogit * max(1.0, float(np.abs(lr).max())), entry
    assert worst <= 2.0 * delta + 1e-12, entry
    if max_rel_rms is not None:
        assert entry["rms_dlogit"] <= max_rel_rms * max(1.0, entry["logit_scale"]), entry
    return entry
