"""TEST INFRASTRUCTURE — CPU (numpy) restatement of the rows either side of the hot path: the input contract
 and the zone_detect output conversion (§8a-12).  Only tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() may import this; the product never does.

Pinning status
  * norm('custom' | 'without'), read_msk one-hot / all-zero -> class 0, parsing_metadata, convert('argmax' |
    'class_prob'): PINNED against outputs of the reference's own functions (tests/golden/make_golden_feed.py ->
    feed_reference.npz, metadata_encoding_reference.json, detect_convert_reference.npz).
  * norm('scaling'): PARITY UNPINNED — scikit-image is absent; restated from its published uint8 -> float64
    conversion (multiply by 1/255 in float64).
  * D4 augmentation: the geometric maps are numpy's flipud / fliplr / rot90 (what albumentations 1.3.1
    VerticalFlip / HorizontalFlip / RandomRotate90 apply); albumentations is absent, so the random draw
    sequence is PARITY UNPINNED.
"""
import numpy as np


def norm_np(img: np.ndarray, norm_type: str, means=(), stds=()) -> np.ndarray:
    """src/flair/data_loader.py:9-30 followed by torch.as_tensor(..., dtype=torch.float) (:86-95)."""
    if norm_type not in ("scaling", "custom", "without"):
        raise SystemExit("Normalization argument should be 'scaling', 'custom' or 'without'.")
    if norm_type == "custom":
        if len(means) != len(stds):
            raise SystemExit("If custom, provided normalization means and stds should be of same lenght.")
        out = img.astype(np.float64)
        for i in range(out.shape[0]):
            out[i] -= means[i]
            out[i] /= stds[i]
        return out.astype(np.float32)
    if norm_type == "scaling":
        return (img.astype(np.float64) * (1.0 / 255.0)).astype(np.float32)
    return img.astype(np.float32)


def onehot_from_raw(raw: np.ndarray, num_classes: int) -> np.ndarray:
    """read_msk, src/flair/data_loader.py:65-69: (raw - 1) in the file's dtype (uint8 wraps 0 -> 255), then a
    stack of equality planes."""
    a = raw - np.asarray(1, dtype=raw.dtype)
    return np.stack([a == i for i in range(num_classes)], axis=0)


def labels_from_raw(raw: np.ndarray, num_classes: int) -> np.ndarray:
    """argmax over the one-hot (task_module.py:71): raw 1..C -> 0..C-1; anything else (0, > C) -> class 0."""
    return np.argmax(onehot_from_raw(raw, num_classes), axis=0).astype(np.uint8)


def d4_np(a: np.ndarray, vflip: bool, hflip: bool, k: int) -> np.ndarray:
    """Compose([VerticalFlip, HorizontalFlip, RandomRotate90]) of src/flair/tasks_utils.py:37-41 for one sample,
    given its draws, on an array whose LAST two axes are (H, W)."""
    if vflip:
        a = a[..., ::-1, :]
    if hflip:
        a = a[..., :, ::-1]
    return np.ascontiguousarray(np.rot90(a, k % 4, axes=(-2, -1)))


def encode_metadata(rec: dict) -> list:
    """One record of parsing_metadata, src/flair/tasks_utils.py:158-213 (including its month term,
    int(month) - 1/12, and the 10e7 scale)."""
    d = 16
    d_i = np.arange(0, d / 2)
    freq = 1 / (10e7 ** (2 * d_i / d))
    x, y = rec["patch_centroid_x"] / 10e7, rec["patch_centroid_y"] / 10e7
    enc = np.zeros(d * 2)
    enc[0:d:2] = np.sin(x * freq)
    enc[1:d:2] = np.cos(x * freq)
    enc[d::2] = np.sin(y * freq)
    enc[d + 1::2] = np.cos(y * freq)
    alti = [(rec["patch_centroid_z"] - 0) / (3164.9099121094 - 0)]
    cam = [1, 0] if "UCE" in rec["camera"] else [0, 1]
    year, month, day = rec["date"].split("-")
    enc_y = {"2018": [1, 0, 0, 0], "2019": [0, 1, 0, 0], "2020": [0, 0, 1, 0], "2021": [0, 0, 0, 1]}[year]
    half = lambda v: (v + 1) / 2  # noqa: E731
    ang_m = 2 * np.pi * (int(month) - 1 / 12)
    ang_d = 2 * np.pi * (int(day) / 31)
    h, m = rec["time"].split("h")
    ang_t = 2 * np.pi * ((int(h) * 3600 + int(m) * 60) / 86400)
    return list(enc) + alti + cam + enc_y + [half(np.sin(ang_m)), half(np.cos(ang_m)), half(np.sin(ang_d)),
                                              half(np.cos(ang_d)), half(np.sin(ang_t)), half(np.cos(ang_t))]


def parsing_metadata_np(image_paths, records: dict) -> list:
    return [encode_metadata(records[p.split("/")[-1][:-4]]) for p in image_paths]


def detect_convert_np(probs: np.ndarray, img_type: str) -> np.ndarray:
    """src/zone_detect/dataset.py:11-34 on a (C, h, w) float32 probability tile."""
    if img_type == "class_prob":
        img = probs
        if img.max() > 1:  # integer inputs only; softmax output never takes this branch
            img = img.astype(np.float32) / np.iinfo(img.dtype).max
        return (img * 255).astype(np.uint8)
    if img_type == "argmax":
        return np.concatenate([np.argmax(probs, axis=0).astype(np.uint8)[None], np.max(probs, axis=0).astype(np.float32)[None]],
                              axis=0)
    return probs


def detect_tile_np(logits: np.ndarray, margin: int, img_type: str) -> np.ndarray:
    """inference + the exact-clipping branch of stitching for one tile: softmax over classes
    (src/zone_detect/compare.py:35), margin crop (:71-75), convert (:76)."""
    x = logits.astype(np.float32)
    e = np.exp(x - x.max(axis=0, keepdims=True))
    p = (e / e.sum(axis=0, keepdims=True)).astype(np.float32)
    S = p.shape[-1]
    p = p[:, margin:S - margin, margin:S - margin]
    return detect_convert_np(p, img_type)
