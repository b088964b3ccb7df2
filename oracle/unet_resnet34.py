"""Oracle (TEST INFRASTRUCTURE): torch-CPU fp32 restatement of smp-0.3.3 ``Unet('resnet34')``.

The reference builds this model through a third-party call,
``smp.create_model(arch, encoder_name, classes, in_channels)``
(/root/reference/src/flair/model.py:37-41); ``segmentation-models-pytorch==0.3.3``
(setup.py:36) and the torchvision ResNet it wraps are not vendored and not
installed, so the published architecture is restated here:

* encoder: torchvision ``resnet34`` without ``fc``/``avgpool``; returns the six
  features ``[x, relu(bn1(conv1 x)), layer1(maxpool .), layer2, layer3, layer4]``;
* decoder: five blocks of nearest x2 upsample, skip concat, 2 x (conv3x3 + BN + ReLU);
* head: conv3x3 16 -> classes with bias.

state_dict keys and shapes follow smp 0.3.3 so the IGNF checkpoints named at
configs/flair-1-config-detect.yaml:13 would load.

PARITY UNPINNED for the arithmetic (no golden outputs exist in the reference);
pinned by known-answer parameter counts only (README.md:91).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out = out + identity
        return self.relu(out)


class ResNet34Encoder(nn.Module):
    """torchvision resnet34 trunk as wrapped by smp ``ResNetEncoder`` (depth 5)."""

    out_channels = (3, 64, 64, 128, 256, 512)

    def __init__(self, in_channels=3):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(64, 3)
        self.layer2 = self._make_layer(128, 4, 2)
        self.layer3 = self._make_layer(256, 6, 2)
        self.layer4 = self._make_layer(512, 3, 2)
        # torchvision ResNet.__init__ initialisation
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._in_channels = 3
        if in_channels != 3:
            self.set_in_channels(in_channels, pretrained=False)

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes, 1, stride, bias=False),
                nn.BatchNorm2d(planes),
            )
        layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(BasicBlock(planes, planes))
        return nn.Sequential(*layers)

    def set_in_channels(self, in_channels, pretrained=True):
        """smp ``EncoderMixin.set_in_channels`` + ``patch_first_conv``."""
        if in_channels == 3:
            return
        self._in_channels = in_channels
        conv = self.conv1
        weight = conv.weight.detach()
        conv.in_channels = in_channels
        if not pretrained:
            conv.weight = nn.Parameter(torch.empty(conv.out_channels, in_channels, *conv.kernel_size))
            conv.reset_parameters()
        elif in_channels == 1:
            conv.weight = nn.Parameter(weight.sum(1, keepdim=True))
        else:
            new_weight = torch.empty(conv.out_channels, in_channels, *conv.kernel_size)
            for i in range(in_channels):
                new_weight[:, i] = weight[:, i % 3]
            conv.weight = nn.Parameter(new_weight * (3 / in_channels))

    def forward(self, x):
        feats = [x]
        x = self.relu(self.bn1(self.conv1(x)))
        feats.append(x)
        x = self.layer1(self.maxpool(x))
        feats.append(x)
        x = self.layer2(x)
        feats.append(x)
        x = self.layer3(x)
        feats.append(x)
        x = self.layer4(x)
        feats.append(x)
        return feats


def _conv2d_relu(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class DecoderBlock(nn.Module):
    def __init__(self, cin, cskip, cout):
        super().__init__()
        self.conv1 = _conv2d_relu(cin + cskip, cout)
        self.attention1 = nn.Identity()
        self.conv2 = _conv2d_relu(cout, cout)
        self.attention2 = nn.Identity()

    def forward(self, x, skip=None):
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        if skip is not None:
            x = torch.cat([x, skip], dim=1)
        return self.conv2(self.conv1(x))


class UnetDecoder(nn.Module):
    def __init__(self, encoder_channels=(3, 64, 64, 128, 256, 512), decoder_channels=(256, 128, 64, 32, 16)):
        super().__init__()
        enc = list(encoder_channels[1:])[::-1]
        in_ch = [enc[0]] + list(decoder_channels[:-1])
        skip_ch = enc[1:] + [0]
        self.center = nn.Identity()
        self.blocks = nn.ModuleList(DecoderBlock(i, s, o) for i, s, o in zip(in_ch, skip_ch, decoder_channels))
        # smp initialize_decoder
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, *features):
        features = features[1:][::-1]
        x = self.center(features[0])
        skips = features[1:]
        for i, blk in enumerate(self.blocks):
            x = blk(x, skips[i] if i < len(skips) else None)
        return x


class Unet(nn.Module):
    """smp-0.3.3 ``Unet(encoder_name='resnet34')``, defaults otherwise."""

    def __init__(self, encoder_name="resnet34", encoder_weights=None, in_channels=3, classes=1):
        super().__init__()
        if encoder_name != "resnet34":
            raise KeyError(f"Wrong encoder name `{encoder_name}`, supported encoders: ['resnet34']")
        self.encoder = ResNet34Encoder(in_channels)
        self.decoder = UnetDecoder()
        self.segmentation_head = nn.Sequential(nn.Conv2d(16, classes, 3, padding=1), nn.Identity(), nn.Identity())
        nn.init.xavier_uniform_(self.segmentation_head[0].weight)
        nn.init.constant_(self.segmentation_head[0].bias, 0)
        self.classification_head = None
        self.name = "u-resnet34"

    def check_input_shape(self, x):
        h, w = x.shape[-2:]
        if h % 32 != 0 or w % 32 != 0:
            nh = (h // 32 + 1) * 32 if h % 32 else h
            nw = (w // 32 + 1) * 32 if w % 32 else w
            raise RuntimeError(
                f"Wrong input shape height={h}, width={w}. Expected image height and width "
                f"divisible by 32. Consider pad your images to shape ({nh}, {nw})."
            )

    def forward(self, x):
        self.check_input_shape(x)
        feats = self.encoder(x)
        return self.segmentation_head(self.decoder(*feats))


def create_model(arch, encoder_name="resnet34", encoder_weights=None, in_channels=3, classes=1, **kwargs):
    """Same signature as ``smp.create_model`` (call site: src/flair/model.py:37-41)."""
    if arch.lower() != "unet":
        raise KeyError(f"Wrong architecture type `{arch}`. Available options are: ['unet']")
    return Unet(encoder_name=encoder_name, encoder_weights=encoder_weights, in_channels=in_channels, classes=classes)


def seeded_model(in_channels=5, classes=13, seed=2022):
    """Deterministic smp-style initialisation from a CPU generator."""
    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        m = create_model("unet", "resnet34", in_channels=in_channels, classes=classes)
    finally:
        torch.random.set_rng_state(state)
    return m
