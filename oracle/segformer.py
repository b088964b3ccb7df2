"""TEST INFRASTRUCTURE — oracle for the SegFormer row (BASELINE config 5).

The reference builds this model through a third-party dependency that is absent from /root/reference:
``transformers.AutoModelForSemanticSegmentation`` (setup.py pins no version; this image ships transformers 5.15.0), see
/root/reference/src/zone_detect/model.py:42-50 and src/flair/model.py:43-50.  The library IS importable here, so the oracle is
the library itself on the CPU in fp32 (``SegformerForSemanticSegmentation`` built from a local ``SegformerConfig``, seeded
random weights — the reference's ``from_pretrained`` is a hub download, impossible without network) plus the x4 bilinear
upsample (align_corners=False) the library applies in front of its own loss.

PARITY UNPINNED against the reference: it holds no test, fixture or output for this path and cannot run it as written
(hub download; logits at 1/4 of the tile size are never rescaled before the margin crop, compare.py:69-75).  Anchors: the
reference's call sites above and the committed fixture tests/golden/segformer_b2_c19.npz made from this oracle by
tests/golden/make_golden_segformer.py.
"""
import torch

MIT_B2 = dict(depths=[3, 4, 6, 3], hidden_sizes=[64, 128, 320, 512], decoder_hidden_size=768, num_attention_heads=[1, 2, 5, 8],
              sr_ratios=[8, 4, 2, 1])


def seeded_model(num_channels=5, num_labels=19, seed=2022, **geometry):
    from transformers import SegformerConfig, SegformerForSemanticSegmentation
    g = dict(MIT_B2)
    g.update(geometry)
    torch.manual_seed(seed)
    m = SegformerForSemanticSegmentation(SegformerConfig(num_channels=num_channels, num_labels=num_labels, **g)).eval()
    # a fresh BatchNorm has mean 0 / variance 1: give the decode head's running statistics something to get wrong
    gen = torch.Generator().manual_seed(seed + 1)
    bn = m.decode_head.batch_norm
    with torch.no_grad():
        bn.running_mean.copy_(0.1 * torch.randn(bn.running_mean.shape, generator=gen))
        bn.running_var.copy_(0.5 + torch.rand(bn.running_var.shape, generator=gen))
        bn.weight.copy_(0.5 + torch.rand(bn.weight.shape, generator=gen))
        bn.bias.copy_(0.1 * torch.randn(bn.bias.shape, generator=gen))
        for p in m.parameters():   # biases and LayerNorm affine parameters are zeros / ones at init: perturb them too
            if p.dim() == 1 and p is not bn.weight and p is not bn.bias:
                p.add_(0.05 * torch.randn(p.shape, generator=gen))
    return m


@torch.no_grad()
def logits(model, x):
    """(quarter-resolution logits = the library's `.logits`, the same after the x4 bilinear upsample)"""
    lq = model(x).logits
    return lq, torch.nn.functional.interpolate(lq, size=x.shape[-2:], mode="bilinear", align_corners=False)
