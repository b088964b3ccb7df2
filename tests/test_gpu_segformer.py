"""SegFormer (MiT-B2, 5 channels, 19 labels: BASELINE config 5) on the HIP executor against the oracle — transformers'
SegformerForSemanticSegmentation on the CPU in fp32 (oracle/segformer.py; parity unpinned against the reference, which holds
nothing for this path) — through the C ABI (flair_segformer_forward).  north_star tolerances: logits within 1e-3 in fp32,
masks by the one parity rule (zero mismatches where the oracle's top-2 probability gap exceeds 1e-5)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(dev, dtype="f32", labels=19, seed=2022, **geom):
    import flair_amd
    from oracle import segformer as osf
    ref = osf.seeded_model(5, labels, seed=seed, **geom)
    kw = {k: geom[k] for k in ("depths", "decoder_hidden_size") if k in geom}
    hip = flair_amd.SegformerForSemanticSegmentation(num_channels=5, num_labels=labels, compute_dtype=dtype, **kw)
    hip.load_state_dict(ref.state_dict(), strict=True)
    return ref, hip.to(dev)


@pytest.mark.parametrize("shape", [(1, 128, 128), (2, 256, 128), (1, 512, 512)])
def test_logits_and_masks_match_transformers_fp32(dev, shape):
    from oracle import parity
    from oracle import segformer as osf
    ref, hip = _pair(dev)
    x = torch.randn(shape[0], 5, shape[1], shape[2], generator=torch.Generator().manual_seed(3))
    lq, lf = osf.logits(ref, x)
    out = hip(x.to(dev))
    gq = out.logits.cpu()
    gf = hip.forward_full(x.to(dev)).cpu()
    assert gq.shape == lq.shape == (shape[0], 19, shape[1] // 4, shape[2] // 4) and gf.shape == lf.shape
    assert float((gq - lq).abs().max()) < 1e-3 and float((gf - lf).abs().max()) < 1e-3
    assert float((gq - lq).abs().max()) < 5e-5   # measured 6e-6: exact-fp32 MFMA chains against MKL's blocking
    parity.assert_mask_parity(f"segformer_b2_{shape[0]}x{shape[1]}x{shape[2]}", lf.argmax(1).numpy(), gf.argmax(1).numpy(),
                              parity.top2_gap(lf.numpy()), logits_ref=lf.numpy(), logits_hip=gf.numpy())


def test_golden_fixture_and_other_geometry(dev, golden_dir):
    """The committed fixture (made from the oracle by tests/golden/make_golden_segformer.py) and a second geometry, MiT-B1
    (depths 2-2-2-2, decode width 256, 13 labels): the executor follows the layer table, not one hard-wired network."""
    from oracle import segformer as osf
    g = np.load(os.path.join(golden_dir, "segformer_b2_c19.npz"))
    _, hip = _pair(dev, seed=int(g["model_seed"]))
    x = torch.randn(1, 5, 256, 256, generator=torch.Generator().manual_seed(int(g["tile_seed"])))
    lq = hip(x.to(dev)).logits.cpu()
    lf = hip.forward_full(x.to(dev)).cpu()
    assert np.abs(lq[0, :, 24:40, 24:40].numpy() - g["logits_quarter_crop"]).max() < 1e-3
    assert np.abs(lq.double().mean(dim=(0, 2, 3)).numpy() - g["logits_quarter_mean"]).max() < 1e-4
    assert np.abs(lf[0, :, 100:132, 100:132].numpy() - g["logits_full_crop"]).max() < 1e-3
    assert float((lf.argmax(1).numpy().astype(np.uint8) != g["mask"]).mean()) < 1e-4   # exact ties only (rule: test above)
    ref, hip1 = _pair(dev, labels=13, seed=5, depths=[2, 2, 2, 2], decoder_hidden_size=256)
    x = torch.randn(2, 5, 128, 128, generator=torch.Generator().manual_seed(4))
    assert float((hip1(x.to(dev)).logits.cpu() - osf.logits(ref, x)[0]).abs().max()) < 5e-5


def test_restructured_decode_head_equals_the_librarys_order(dev):
    """The decode head runs its products at each stage's own resolution with pre-multiplied weights (linearity: a per-channel
    bilinear resize commutes with a per-pixel channel mix; csrc/segformer_ops.hip).  FLAIR_SF_HEAD=0 runs the library's order
    (Linear, upsample, concatenate, 1x1 fuse): same logits up to fp32 summation order, and both sit on the oracle."""
    from flair_amd import _lib as L
    from oracle import segformer as osf
    ref, hip = _pair(dev)
    x = torch.randn(2, 5, 256, 256, generator=torch.Generator().manual_seed(9))
    want = osf.logits(ref, x)[0]
    try:
        L.check(L.lib().flair_tune_set(b"FLAIR_SF_HEAD", 0))
        a = hip(x.to(dev)).logits.cpu()
    finally:
        L.lib().flair_tune_set(b"FLAIR_SF_HEAD", 2)
    b = hip(x.to(dev)).logits.cpu()
    assert float((a - b).abs().max()) < 2e-5
    assert float((a - want).abs().max()) < 5e-5 and float((b - want).abs().max()) < 5e-5


@pytest.mark.parametrize("shape", [(2, 256, 128), (1, 128, 128), (1, 512, 512)])
def test_fused_decode_head_kernel_matches_the_separate_passes_bf16(dev, shape):
    """bf16 runs everything after the per-stage products of the decode head in one kernel (head_fused_kernel: stage-0 product,
    the three bilinear upsamples as a product with a constant interpolation matrix, folded BatchNorm + ReLU, classifier; 16 x 8
    pixel tiles).  FLAIR_SF_HEAD=1 runs the separate passes (product -> upsample-sum -> classifier, each rounding its output to
    bf16).  The two differ by those roundings only: measured max 4e-3 / rms 6e-4 of the logit scale; edge tiles (clamped source
    pixels), a non-square grid and the 2 x 2-tile minimum are covered by the shapes.  Both sit on the fp32 oracle by the bf16 rule."""
    from flair_amd import _lib as L
    from oracle import parity
    from oracle import segformer as osf
    ref, hip = _pair(dev, "bf16")
    x = torch.randn(*[shape[0], 5, shape[1], shape[2]], generator=torch.Generator().manual_seed(21))
    fused = hip(x.to(dev)).logits.cpu()
    try:
        L.check(L.lib().flair_tune_set(b"FLAIR_SF_HEAD", 1))
        sep = hip(x.to(dev)).logits.cpu()
    finally:
        L.lib().flair_tune_set(b"FLAIR_SF_HEAD", 2)
    want = osf.logits(ref, x)[0]
    scale = float(want.abs().max())
    d = (fused - sep).abs()
    parity.record({"test": f"segformer_fused_head_{shape[0]}x{shape[1]}x{shape[2]}", "logit_scale": scale, "max_abs": float(d.max()),
                   "rms": float(d.pow(2).mean().sqrt()), "fused_vs_oracle_max": float((fused - want).abs().max()),
                   "separate_vs_oracle_max": float((sep - want).abs().max())})
    assert float(d.max()) < 1.5e-2 * scale and float(d.pow(2).mean().sqrt()) < 2e-3 * scale
    assert float((fused - want).abs().max()) < 3e-2 * scale and float((fused - want).pow(2).mean().sqrt()) < 6e-3 * scale


@pytest.mark.parametrize("shape", [(2, 256, 128), (1, 128, 128), (1, 512, 512)])
def test_fused_mix_ffn_kernel_matches_the_separate_passes_bf16(dev, shape):
    """bf16 runs LayerNorm -> fc1 -> depth-wise 3x3 + GELU -> fc2 -> + residual of the 64- and 128-channel stages in one kernel per
    block (ffn_fused_kernel: 8 x 8 token tiles with a recomputed one-pixel halo, the 4C-wide intermediate stays in LDS).
    FLAIR_SF_FFN=0 runs the four separate passes.  Both round the same intermediates to bf16 (LayerNorm output, fc1 output, GELU
    output, the block's result), so they differ by summation order and the bf16 roundings it flips, through 7 blocks; image-edge
    tiles (zero padding of the depth-wise conv's input), a non-square grid and the 128 x 128 minimum are in the shapes."""
    from flair_amd import _lib as L
    from oracle import parity
    from oracle import segformer as osf
    ref, hip = _pair(dev, "bf16")
    x = torch.randn(*[shape[0], 5, shape[1], shape[2]], generator=torch.Generator().manual_seed(22))
    fused = hip(x.to(dev)).logits.cpu()
    try:
        L.check(L.lib().flair_tune_set(b"FLAIR_SF_FFN", 0))
        sep = hip(x.to(dev)).logits.cpu()
    finally:
        L.lib().flair_tune_set(b"FLAIR_SF_FFN", 1)
    want = osf.logits(ref, x)[0]
    scale = float(want.abs().max())
    d = (fused - sep).abs()
    parity.record({"test": f"segformer_fused_ffn_{shape[0]}x{shape[1]}x{shape[2]}", "logit_scale": scale, "max_abs": float(d.max()),
                   "rms": float(d.pow(2).mean().sqrt()), "fused_vs_oracle_max": float((fused - want).abs().max()),
                   "separate_vs_oracle_max": float((sep - want).abs().max())})
    assert float(d.max()) < 2e-2 * scale and float(d.pow(2).mean().sqrt()) < 3e-3 * scale
    assert float((fused - want).abs().max()) < 3e-2 * scale and float((fused - want).pow(2).mean().sqrt()) < 6e-3 * scale


def test_bf16_shapes_the_fused_kernels_do_not_take(dev):
    """512 x 96 (16 x 3 = 48 reduced keys) is a valid tile whose 1/4 grid is 24 wide — not a whole number of the fused decode
    head's 16-pixel tiles — and whose stage-1 grid is 12 wide — not a whole number of the fused Mix-FFN's 8 x 8 tiles: those two
    fall back to the separate passes (stage 0 stays fused), batch 3.  Same bf16 rule as below."""
    from oracle import parity
    from oracle import segformer as osf
    ref, hip = _pair(dev, "bf16")
    x = torch.randn(3, 5, 512, 96, generator=torch.Generator().manual_seed(5))
    _, lf = osf.logits(ref, x)
    gf = hip.forward_full(x.to(dev)).cpu()
    parity.assert_masks_within_logit_error("segformer_b2_bf16_3x512x96", lf.numpy(), gf.numpy(), gf.argmax(1).numpy(),
                                           max_rel_dlogit=3e-2, max_rel_rms=6e-3)


@pytest.mark.parametrize("side", [512, 384])
def test_bf16_mode_tracks_the_oracle(dev, side):
    """bf16 throughput mode: the logit-error rule of oracle/parity.py (measured: max |dlogit| 1.0e-2 and rms 1.9e-3 of the
    logit scale at 512x512, profiles/r3_parity.json; bounds 3x).  384 x 384 has 144 reduced keys: the attention kernel's second
    half of keys (online softmax across halves of 128) holds a single 16-key tile."""
    from oracle import parity
    from oracle import segformer as osf
    ref, hip = _pair(dev, "bf16")
    x = torch.randn(1, 5, side, side, generator=torch.Generator().manual_seed(3))
    _, lf = osf.logits(ref, x)
    gf = hip.forward_full(x.to(dev)).cpu()
    parity.assert_masks_within_logit_error(f"segformer_b2_bf16_1x{side}", lf.numpy(), gf.numpy(), gf.argmax(1).numpy(),
                                           max_rel_dlogit=3e-2, max_rel_rms=6e-3)


def test_two_tile_attention_kernel_matches_the_one_tile_form_bf16(dev):
    """attention2_kernel (two query tiles per wave, keys in halves of 128 with an online softmax) against attention_kernel (all
    keys at once), FLAIR_SF_ATT2 = 1 / 0, on 16, 64, 144 and 256 reduced keys: same products, different summation order of the
    softmax denominator and one more rescale — bf16 rounding level."""
    from flair_amd import _lib as L
    _, hip = _pair(dev, "bf16")
    for side in (128, 256, 384, 512):
        x = torch.randn(1, 5, side, side, generator=torch.Generator().manual_seed(side))
        a = hip(x.to(dev)).logits.cpu()
        try:
            L.check(L.lib().flair_tune_set(b"FLAIR_SF_ATT2", 0))
            b = hip(x.to(dev)).logits.cpu()
        finally:
            L.lib().flair_tune_set(b"FLAIR_SF_ATT2", 1)
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) < 1e-2 * scale and float((a - b).pow(2).mean().sqrt()) < 2e-3 * scale, side


def test_contract(dev):
    import flair_amd
    from flair_amd._lib import FlairHipError
    m = flair_amd.SegformerForSemanticSegmentation(num_channels=5, num_labels=19, depths=(1, 1, 1, 1))
    with pytest.raises(FlairHipError):
        m(torch.zeros(1, 5, 128, 128))                  # host tensors are refused: no CPU fallback
    m = m.to(dev)
    with pytest.raises(RuntimeError, match="multiples of 32"):
        m(torch.zeros(1, 5, 96, 96, device=dev))        # (3 x 3 = 9 reduced tokens: not a whole 16-key tile)
    with pytest.raises(RuntimeError):
        m.train()
    assert m(torch.zeros(2, 5, 128, 128, device=dev)).logits.shape == (2, 19, 32, 32)
    # the factory's HuggingFace provider (model.py:43-50, 66-68)
    cfg = {"model_framework": {"model_provider": "HuggingFace", "HuggingFace": {"org_model": "nvidia/mit-b2"}},
           "use_metadata": False, "channels": [1, 2, 3, 4, 5], "classes": {i: [1, str(i)] for i in range(1, 20)}}
    f = flair_amd.FLAIR_ModelFactory(cfg).to(dev)
    assert f(torch.zeros(1, 5, 128, 128, device=dev)).shape == (1, 19, 32, 32)
    with pytest.raises(NotImplementedError):
        flair_amd.FLAIR_ModelFactory({**cfg, "model_framework": {"model_provider": "HuggingFace",
                                                                 "HuggingFace": {"org_model": "openmmlab/upernet-swin-small"}}})


def test_cached_weight_layouts_follow_the_parameters(dev):
    """The library keeps packed weights / the decode head's pre-multiplied matrices between forwards
    (flair_segformer_weights_changed): an in-place update of the parameters — load_state_dict, copy_ — must be seen by the next
    forward, the same weights must give the same bits with a warm and a cold cache, and another batch size (another workspace)
    must not read stale layouts."""
    from oracle import segformer as osf
    ref, hip = _pair(dev, seed=11)
    x = torch.randn(2, 5, 128, 128, generator=torch.Generator().manual_seed(8))
    cold = hip(x.to(dev)).logits.clone()
    warm = hip(x.to(dev)).logits.clone()
    assert torch.equal(cold, warm)
    ref2 = osf.seeded_model(5, 19, seed=12)
    hip.load_state_dict(ref2.state_dict(), strict=True)               # in place: same pointers, new contents
    assert float((hip(x.to(dev)).logits.cpu() - osf.logits(ref2, x)[0]).abs().max()) < 5e-5
    with torch.no_grad():
        hip.decode_head.classifier.bias.add_(1.0)                     # a single tensor, in place
    assert float((hip(x.to(dev)).logits.cpu() - (osf.logits(ref2, x)[0] + 1.0)).abs().max()) < 5e-5
    big = torch.randn(4, 5, 256, 256, generator=torch.Generator().manual_seed(9))   # larger workspace -> new buffer
    with torch.no_grad():
        ref2.decode_head.classifier.bias.add_(1.0)
    assert float((hip(big.to(dev)).logits.cpu() - osf.logits(ref2, big)[0]).abs().max()) < 5e-5
    assert float((hip(x.to(dev)).logits.cpu() - osf.logits(ref2, x)[0]).abs().max()) < 5e-5


def test_zone_detector_with_segformer_matches_sequential_replay(dev):
    """zone_detect's window loop (slicing -> model -> softmax -> margin crop -> convert -> write, compare.py:20-39,69-82) with
    the HuggingFace provider, on a raster that is not a multiple of the stride: the device pipeline against the sequential CPU
    restatement (oracle/zone_detect.py) fed the oracle's logits upsampled x4 to the tile size."""
    from flair_amd.zone_detect import ZoneDetector
    from oracle import parity
    from oracle import segformer as osf
    from oracle import zone_detect as oz
    ref, hip = _pair(dev, depths=[1, 1, 1, 1], decoder_hidden_size=256)

    class _Upsampled:   # what compare.py needs from a model whose logits come at 1/4 resolution
        def eval(self):
            return self

        def __call__(self, x):
            return osf.logits(ref, x)[1]

    cfg = {"img_pixels_detection": 128, "margin": 32, "output_type": "argmax", "n_classes": 19, "batch_size": 3,
           "channels": [1, 2, 3, 4, 5], "norma_task": [{"norm_type": "custom", "norm_means": [105.08, 110.87, 101.82, 106.38, 53.26],
                                                      "norm_stds": [52.17, 45.38, 44, 39.69, 79.3]}]}
    raster = np.random.default_rng(4).integers(0, 256, size=(5, 200, 264), dtype=np.uint8)
    gap = np.zeros((200, 264))
    want = oz.detect_raster_np(_Upsampled(), raster, cfg, gap_out=gap)
    got = ZoneDetector(hip, cfg).run(torch.from_numpy(raster).to(dev)).cpu().numpy()
    assert got.shape == want.shape == (2, 200, 264)
    parity.assert_mask_parity("zone_detector_segformer_200x264", want[0], got[0], gap)
    assert np.abs(got[1] - want[1]).max() < 1e-4
    cfg2 = dict(cfg, output_type="class_prob", batch_size=5)
    want = oz.detect_raster_np(_Upsampled(), raster, cfg2)
    got = ZoneDetector(hip, cfg2).run(torch.from_numpy(raster).to(dev)).cpu().numpy()
    assert got.shape == (19, 200, 264) and got.dtype == np.uint8
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
