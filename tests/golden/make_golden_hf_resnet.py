"""Second source for the ResNet34 trunk of the oracle: the smp / torchvision encoder is absent from
/root/reference and from this image, but `transformers` ships an independent implementation of the same network,
ResNetModel(ResNetConfig(layer_type="basic", depths=[3, 4, 6, 3], hidden_sizes=[64, 128, 256, 512])).  This script loads
the ORACLE's seeded weights into it (pure key remap, no arithmetic), runs both BatchNorm modes on a seeded 2x5x64x64 input
and stores the five feature maps.  tests/test_oracle_cpu.py then checks oracle/unet_resnet34.py's encoder against the
file (and, where transformers is importable, against the live model).  Run in the build container:

    python tests/golden/make_golden_hf_resnet.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def hf_resnet34(in_channels):
    from transformers import ResNetConfig, ResNetModel
    cfg = ResNetConfig(num_channels=in_channels, embedding_size=64, hidden_sizes=[64, 128, 256, 512], depths=[3, 4, 6, 3],
                       layer_type="basic", hidden_act="relu", downsample_in_first_stage=False)
    return ResNetModel(cfg)


def remap(oracle_sd):
    """torchvision/smp encoder key -> transformers key (state only moves, nothing is computed)."""
    out = {}
    for k, v in oracle_sd.items():
        if not k.startswith("encoder."):
            continue
        p = k[len("encoder."):].split(".")
        if p[0] == "conv1":
            nk = "embedder.embedder.convolution." + p[1]
        elif p[0] == "bn1":
            nk = "embedder.embedder.normalization." + p[1]
        else:
            stage, blk = int(p[0][len("layer"):]) - 1, int(p[1])
            base = f"encoder.stages.{stage}.layers.{blk}."
            if p[2] in ("conv1", "conv2"):
                nk = base + f"layer.{int(p[2][-1]) - 1}.convolution." + p[3]
            elif p[2] in ("bn1", "bn2"):
                nk = base + f"layer.{int(p[2][-1]) - 1}.normalization." + p[3]
            else:   # downsample.0 = conv, downsample.1 = bn
                nk = base + ("shortcut.convolution." if p[3] == "0" else "shortcut.normalization.") + p[4]
        out[nk] = v
    return out


def hf_features(model, x):
    """[stem (pre-pool), stage1..4] — the tensors smp's ResNetEncoder.forward returns after the identity."""
    stem = model.embedder.embedder(x)
    out = model(x, output_hidden_states=True)
    return [stem] + list(out.hidden_states[1:])


def main():
    from oracle import unet_resnet34 as om
    ref = om.seeded_model(5, 13, seed=2022)
    hf = hf_resnet34(5)
    missing, unexpected = hf.load_state_dict(remap(ref.state_dict()), strict=False)
    assert not unexpected and all("num_batches_tracked" in m for m in missing), (missing, unexpected)
    x = torch.randn(2, 5, 64, 64, generator=torch.Generator().manual_seed(7))
    data = {"x": x.numpy(), "seed_model": 2022, "seed_x": 7}
    with torch.no_grad():
        hf.eval()
        for i, f in enumerate(hf_features(hf, x)):
            data[f"eval_f{i + 1}"] = f.numpy()
        hf.train()
        for i, f in enumerate(hf_features(hf, x)):
            data[f"train_f{i + 1}"] = f.numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hf_resnet34_features.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, {k: v.shape for k, v in data.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main()
