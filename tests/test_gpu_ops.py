"""Operator-level parity: HIP kernels (through the C ABI) vs torch-CPU fp32 on the same seeded inputs.

Tolerances: fp32 mode (exact-fp32 MFMA, different summation order than MKL-DNN) 2e-4 relative to the
tensor's max magnitude; bf16 mode 3e-2 (8-bit mantissa inputs, fp32 accumulation)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 2e-4, torch.bfloat16: 3e-2}


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _nhwc(x, dtype, dev):
    return x.permute(0, 2, 3, 1).contiguous().to(device=dev, dtype=dtype)


def _q(x, dtype):  # quantise inputs to the compute dtype so the reference sees the same values
    return x.to(dtype).float()


CONV_CASES = [
    # N, H, W, Cin, Cout, R, stride, pad
    (2, 16, 16, 64, 64, 3, 1, 1),
    (1, 24, 40, 64, 128, 3, 2, 1),
    (2, 8, 8, 256, 256, 3, 1, 1),
    (3, 8, 8, 128, 256, 1, 2, 0),
    (2, 32, 32, 8, 64, 7, 2, 3),
    (3, 64, 96, 8, 64, 7, 2, 3),
    (1, 32, 32, 32, 16, 3, 1, 1),
    (1, 32, 32, 128, 32, 3, 1, 1),
    (2, 4, 4, 512, 512, 3, 1, 1),
    (1, 2, 2, 256, 512, 3, 2, 1),
    # halo-GEMM kernel (3x3 s1, >= 64 channels, 8x32 / 16x16 tiles)
    (1, 8, 32, 64, 64, 3, 1, 1),
    (2, 16, 16, 128, 128, 3, 1, 1),
    (1, 16, 64, 128, 256, 3, 1, 1),
    (1, 8, 64, 64, 192, 3, 1, 1),
    (1, 32, 16, 512, 128, 3, 1, 1),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward_and_stats(dev, dtype, case):
    from flair_amd import ops
    N, H, W, Cin, Cout, R, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = _q(torch.randn(N, Cin, H, W, generator=g), dtype)
    w = torch.randn(Cout, Cin, R, R, generator=g) / (Cin * R * R) ** 0.5
    ref = F.conv2d(x, _q(w, dtype), stride=stride, padding=pad)
    y, yn, st = ops.conv2d_forward(_nhwc(x, dtype, dev), w.to(dev), stride=stride, pad=pad, want_nchw=True, want_stats=True)
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert _rel(got, ref) < TOL[dtype]
    assert _rel(yn.cpu(), ref) < TOL[dtype]
    s1, s2 = got.sum(dim=(0, 2, 3)), (got * got).sum(dim=(0, 2, 3))
    assert _rel(st[0].cpu(), s1) < 1e-4 and _rel(st[1].cpu(), s2) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_upsample_concat_bias_head(dev, dtype):
    """decoder conv1 gather (nearest x2 upsample of x0 ++ skip) and the biased 13-class head (NCHW fp32 out)."""
    from flair_amd import ops
    g = torch.Generator().manual_seed(3)
    x0 = _q(torch.randn(2, 64, 8, 8, generator=g), dtype)
    x1 = _q(torch.randn(2, 32, 16, 16, generator=g), dtype)
    w = torch.randn(32, 96, 3, 3, generator=g) / 30
    ref = F.conv2d(torch.cat([F.interpolate(x0, scale_factor=2, mode="nearest"), x1], 1), _q(w, dtype), padding=1)
    y, _, _ = ops.conv2d_forward(_nhwc(x0, dtype, dev), w.to(dev), x1=_nhwc(x1, dtype, dev), up0=True)
    assert _rel(y.float().cpu().permute(0, 3, 1, 2), ref) < TOL[dtype]
    # halo-GEMM path with two sources: 128 upsampled + 64 skip channels -> 128 (decoder block 2 shape)
    x0 = _q(torch.randn(1, 128, 8, 16, generator=g), dtype)
    x1 = _q(torch.randn(1, 64, 16, 32, generator=g), dtype)
    w = torch.randn(128, 192, 3, 3, generator=g) / 42
    ref = F.conv2d(torch.cat([F.interpolate(x0, scale_factor=2, mode="nearest"), x1], 1), _q(w, dtype), padding=1)
    y, _, st = ops.conv2d_forward(_nhwc(x0, dtype, dev), w.to(dev), x1=_nhwc(x1, dtype, dev), up0=True, want_stats=True)
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert _rel(got, ref) < TOL[dtype]
    assert _rel(st[0].cpu(), got.sum(dim=(0, 2, 3))) < 1e-4
    # decoder block 3 conv1 shape (64 upsampled + 64 skip -> 32): the 32-wide column block of the LDS-DMA halo-GEMM,
    # on both of its tile shapes (32x8 and 16x16)
    for hh, ww in ((8, 32), (16, 16)):
        x0 = _q(torch.randn(2, 64, hh // 2, ww // 2, generator=g), dtype)
        x1 = _q(torch.randn(2, 64, hh, ww, generator=g), dtype)
        w = torch.randn(32, 128, 3, 3, generator=g) / 34
        ref = F.conv2d(torch.cat([F.interpolate(x0, scale_factor=2, mode="nearest"), x1], 1), _q(w, dtype), padding=1)
        y, _, st = ops.conv2d_forward(_nhwc(x0, dtype, dev), w.to(dev), x1=_nhwc(x1, dtype, dev), up0=True, want_stats=True)
        got = y.float().cpu().permute(0, 3, 1, 2)
        assert _rel(got, ref) < TOL[dtype]
        assert _rel(st[0].cpu(), got.sum(dim=(0, 2, 3))) < 1e-4 and _rel(st[1].cpu(), (got * got).sum(dim=(0, 2, 3))) < 1e-4
    xh = _q(torch.randn(2, 16, 32, 32, generator=g), dtype)
    wh = torch.randn(13, 16, 3, 3, generator=g) / 12
    bh = torch.randn(13, generator=g)
    refh = F.conv2d(xh, _q(wh, dtype), bh, padding=1)
    y, yn, _ = ops.conv2d_forward(_nhwc(xh, dtype, dev), wh.to(dev), bias=bh.to(dev), want_nchw=True)
    assert y is None and _rel(yn.cpu(), refh) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[5] != 7])
def test_conv_backward(dev, dtype, case):
    from flair_amd import ops
    N, H, W, Cin, Cout, R, stride, pad = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = _q(torch.randn(N, Cin, H, W, generator=g), dtype).requires_grad_(True)
    w = _q(torch.randn(Cout, Cin, R, R, generator=g) / (Cin * R * R) ** 0.5, dtype).requires_grad_(True)
    y = F.conv2d(x, w, stride=stride, padding=pad)
    dy = _q(torch.randn(y.shape, generator=g), dtype)
    y.backward(dy)
    dx, dw = ops.conv2d_backward(_nhwc(x.detach(), dtype, dev), w.detach().to(dev), _nhwc(dy, dtype, dev), stride=stride, pad=pad)
    assert _rel(dx.float().cpu().permute(0, 3, 1, 2), x.grad) < TOL[dtype]
    assert _rel(dw.cpu(), w.grad) < TOL[dtype]


def test_stem_wgrad_padded_channels(dev):
    """7x7 s2 stem: 5 real channels stored as 8; dw keeps the real Cin."""
    import ctypes as C
    from flair_amd import _lib as L, ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 5, 32, 32, generator=g).requires_grad_(True)
    w = (torch.randn(64, 5, 7, 7, generator=g) / 15).requires_grad_(True)
    y = F.conv2d(x, w, stride=2, padding=3)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    x8 = ops.nchw_to_nhwc(x.detach().to(dev), torch.float32, 8)
    w8 = torch.zeros(64, 8, 7, 7)
    w8[:, :5] = w.detach()
    yk, _, _ = ops.conv2d_forward(x8, w8.to(dev), stride=2, pad=3)
    assert _rel(yk.cpu().permute(0, 3, 1, 2), y.detach()) < 2e-4
    _, dw8 = ops.conv2d_backward(x8, w8.to(dev), _nhwc(dy, torch.float32, dev), stride=2, pad=3, need_dx=False)
    assert _rel(dw8.cpu()[:, :5], w.grad) < 2e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,relu,res", [(64, True, False), (16, True, True), (512, False, False), (128, True, True)])
def test_bn_relu_forward_backward(dev, dtype, C, relu, res):
    from flair_amd import ops
    g = torch.Generator().manual_seed(C)
    N, H, W = 3, 10, 12
    y = _q(torch.randn(N, C, H, W, generator=g) * 2 + 0.5, dtype).requires_grad_(True)
    r = _q(torch.randn(N, C, H, W, generator=g), dtype).requires_grad_(True) if res else None
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.1)
    rm, rv = bn.running_mean.clone().to(dev), bn.running_var.clone().to(dev)
    o = bn(y)
    if res:
        o = o + r
    if relu:
        o = torch.relu(o)
    do = _q(torch.randn(o.shape, generator=g), dtype)
    o.backward(do)
    out, mean, invstd = ops.bn_relu_forward(_nhwc(y.detach(), dtype, dev), bn.weight.detach().to(dev), bn.bias.detach().to(dev),
                                            rm, rv, training=True, residual=_nhwc(r.detach(), dtype, dev) if res else None, relu=relu)
    tol = TOL[dtype]
    assert _rel(out.float().cpu().permute(0, 3, 1, 2), o.detach()) < tol
    assert _rel(rm.cpu(), bn.running_mean) < 1e-5 and _rel(rv.cpu(), bn.running_var) < 1e-5
    out_ref = _nhwc(o.detach(), dtype, dev)  # use the reference activation for the ReLU mask
    dy, dres, dg, db = ops.bn_relu_backward(_nhwc(do, dtype, dev), out_ref, _nhwc(y.detach(), dtype, dev), bn.weight.detach().to(dev),
                                            mean, invstd, relu=relu, want_dres=res)
    assert _rel(dy.float().cpu().permute(0, 3, 1, 2), y.grad) < tol
    assert _rel(dg.cpu(), bn.weight.grad) < tol and _rel(db.cpu(), bn.bias.grad) < tol
    if res:
        assert _rel(dres.float().cpu().permute(0, 3, 1, 2), r.grad) < tol
    # eval mode uses the running statistics
    bn.eval()
    oe = torch.relu(bn(y.detach())) if relu else bn(y.detach())
    oute, _, _ = ops.bn_relu_forward(_nhwc(y.detach(), dtype, dev), bn.weight.detach().to(dev), bn.bias.detach().to(dev),
                                     bn.running_mean.to(dev), bn.running_var.to(dev), training=False, relu=relu)
    assert _rel(oute.float().cpu().permute(0, 3, 1, 2), oe) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_maxpool(dev, dtype):
    from flair_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.relu(_q(torch.randn(2, 64, 16, 24, generator=g), dtype))  # ReLU output: many exact ties at 0
    x.requires_grad_(True)
    y = F.max_pool2d(x, 3, 2, 1)
    dy = _q(torch.randn(y.shape, generator=g), dtype)
    y.backward(dy)
    yk, idx = ops.maxpool_forward(_nhwc(x.detach(), dtype, dev))
    assert torch.equal(yk.float().cpu().permute(0, 3, 1, 2), y.detach())
    dx = ops.maxpool_backward(_nhwc(dy, dtype, dev), idx, 16, 24)
    assert _rel(dx.float().cpu().permute(0, 3, 1, 2), x.grad) < TOL[dtype]


@pytest.mark.parametrize("C", [13, 19])
@pytest.mark.parametrize("kind", ["u8", "i64", "onehot"])
def test_ce_head_against_oracle(dev, C, kind):
    """Fused head vs the numpy/torch oracle (oracle/seg_step.py): loss, dlogits, preds, targets, confmat."""
    from flair_amd import ops
    from oracle import seg_step
    g = torch.Generator().manual_seed(C)
    B, H, W = 2, 24, 40
    logits = torch.randn(B, C, H, W, generator=g) * 3
    lab = torch.randint(0, C, (B, H, W), generator=g)
    w = torch.rand(C, generator=g) + 0.1
    w[2] = 0.0
    onehot = torch.stack([(lab == i) for i in range(C)], 1).float()
    onehot[0, :, :3, :3] = 0  # all-zero one-hot pixels -> class 0 (task_module.py:71)
    lab_eff = onehot.argmax(1)
    loss_ref, preds_ref, targets_ref = seg_step.step_torch(logits.clone().requires_grad_(True), onehot, w)
    loss_np, dl_np = seg_step.cross_entropy_np(logits.numpy(), lab_eff.numpy(), w.numpy())
    labels = {"u8": lab_eff.to(torch.uint8), "i64": lab_eff, "onehot": onehot}[kind].to(dev)
    cm = torch.zeros(C, C, dtype=torch.int64, device=dev)
    loss, dl, preds, tg = ops.ce_head(logits.to(dev), labels, w.to(dev), want_preds="i64", confmat=cm, want_targets=True)
    assert abs(loss.item() - loss_np) < 2e-6 * max(1, abs(loss_np)) and abs(loss.item() - loss_ref.item()) < 1e-5
    assert np.abs(dl.cpu().numpy() - dl_np).max() < 1e-9 + 1e-5 * np.abs(dl_np).max()
    assert torch.equal(preds.cpu().flatten(1), preds_ref)
    assert torch.equal(tg.cpu().flatten(1), targets_ref)
    cm_ref = seg_step.confusion_matrix_np(targets_ref.numpy(), preds_ref.numpy(), C)
    assert np.array_equal(cm.cpu().numpy(), cm_ref)
    per, wj, mj = ops.jaccard(cm)
    assert np.allclose(per.cpu().numpy(), seg_step.jaccard_from_confmat(cm_ref, None), atol=1e-6)
    assert abs(wj.item() - seg_step.jaccard_from_confmat(cm_ref, "weighted")) < 1e-6
    assert abs(mj.item() - seg_step.jaccard_from_confmat(cm_ref, "macro")) < 1e-6


def test_confmat_and_softmax_argmax(dev, golden_dir):
    import json, os
    from flair_amd import ops
    from oracle import seg_step
    gold = json.load(open(os.path.join(golden_dir, "jaccard_restated.json")))
    t = torch.tensor(gold["target"], dtype=torch.int32, device=dev)
    p = torch.tensor(gold["pred"], dtype=torch.int64, device=dev)
    p[5] = 40  # out-of-range pair is dropped (sklearn labels=range(C) semantics, metrics.py:67-71)
    cm = ops.confmat_update(torch.zeros(13, 13, dtype=torch.int64, device=dev), t, p)
    ref = np.array(gold["confmat"])
    ref[gold["target"][5], gold["pred"][5]] -= 1
    assert np.array_equal(cm.cpu().numpy(), ref)
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(2, 19, 16, 16, generator=g) * 4
    logits[0, 3, 0, 0] = logits[0, 7, 0, 0] = 50.0  # exact tie -> first index
    pr, mp = ops.softmax_argmax(logits.to(dev), want="u8", want_maxprob=True)
    probs = torch.softmax(logits, 1)
    assert torch.equal(pr.cpu().long(), seg_step.predict_torch(logits))
    conv = seg_step.detect_convert_argmax(probs[0].numpy())
    assert np.array_equal(conv[0], pr[0].cpu().numpy().astype(np.float32))
    assert np.abs(conv[1] - mp[0].cpu().numpy()).max() < 1e-6


HALO_CASES = [
    # N, H, W, C0, C1, up0, Cout  (3x3 s1 p1; served by the halo-tile kernel: H % 8 == 0, W % 32 == 0)
    (2, 16, 64, 16, 0, False, 16),
    (1, 24, 32, 16, 0, False, 13),
    (2, 16, 32, 32, 0, True, 16),
    (1, 16, 64, 32, 0, False, 32),
    (1, 16, 32, 64, 64, True, 32),
    (2, 8, 32, 32, 0, False, 128),
    (1, 8, 64, 16, 0, False, 32),
    (2, 16, 32, 64, 0, False, 64),
    (1, 16, 32, 128, 0, False, 32),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", HALO_CASES)
def test_conv_halo_kernel(dev, dtype, case):
    """Small-channel 3x3 layers of the decoder tail / head (fwd, data gradient, stats, bias, NCHW logits)."""
    from flair_amd import ops
    N, H, W, C0, C1, up0, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    h0, w0 = (H // 2, W // 2) if up0 else (H, W)
    x0 = _q(torch.randn(N, C0, h0, w0, generator=g), dtype)
    x1 = _q(torch.randn(N, C1, H, W, generator=g), dtype) if C1 else None
    w = _q(torch.randn(Cout, C0 + C1, 3, 3, generator=g) / (9 * (C0 + C1)) ** 0.5, dtype).requires_grad_(True)
    b = torch.randn(Cout, generator=g) if Cout == 13 else None
    xin = F.interpolate(x0, scale_factor=2, mode="nearest") if up0 else x0
    if C1:
        xin = torch.cat([xin, x1], 1)
    xin.requires_grad_(True)
    ref = F.conv2d(xin, w, b, padding=1)
    wd = w.detach().to(dev)
    y, yn, st = ops.conv2d_forward(_nhwc(x0, dtype, dev), wd, bias=None if b is None else b.to(dev), x1=None if x1 is None else _nhwc(x1, dtype, dev),
                                   up0=up0, want_nchw=True, want_stats=b is None)
    assert _rel(yn.cpu(), ref.detach()) < TOL[dtype]
    if y is not None:
        got = y.float().cpu().permute(0, 3, 1, 2)
        assert _rel(got, ref.detach()) < TOL[dtype]
        assert _rel(st[0].cpu(), got.sum(dim=(0, 2, 3))) < 1e-4 and _rel(st[1].cpu(), (got * got).sum(dim=(0, 2, 3))) < 1e-4
    if not up0 and not C1 and Cout % 8 == 0:
        dy = _q(torch.randn(ref.shape, generator=g), dtype)
        ref.backward(dy)
        dx, dw = ops.conv2d_backward(_nhwc(x0, dtype, dev), wd, _nhwc(dy, dtype, dev))
        assert _rel(dx.float().cpu().permute(0, 3, 1, 2), xin.grad) < TOL[dtype]
        assert _rel(dw.cpu(), w.grad) < TOL[dtype]  # halo-tile weight gradient (all nine taps per workgroup)


@pytest.mark.parametrize("shape", [(2, 32, 32), (3, 64, 96), (1, 128, 128)])
def test_stem_direct_kernels_bf16(dev, shape):
    """The 7x7 stride-2 stem in bf16 takes the dedicated direct kernels (stem.hip) when the output is NHWC only:
    forward + BatchNorm partial statistics and the weight gradient against torch on the same rounded operands."""
    from flair_amd import ops
    N, H, W = shape
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(H + W)
    x = _q(torch.randn(N, 8, H, W, generator=g), dtype)
    x[:, 5:] = 0  # channels 5..7 are padding in the network
    w = _q(torch.randn(64, 8, 7, 7, generator=g) / (5 * 49) ** 0.5, dtype).requires_grad_(True)
    ref = F.conv2d(x, w, stride=2, padding=3)
    y, _, st = ops.conv2d_forward(_nhwc(x, dtype, dev), w.detach().to(dev), stride=2, pad=3, want_stats=True)
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert _rel(got, ref.detach()) < TOL[dtype]
    assert _rel(st[0].cpu(), got.sum(dim=(0, 2, 3))) < 1e-4 and _rel(st[1].cpu(), (got * got).sum(dim=(0, 2, 3))) < 1e-4
    dy = _q(torch.randn(ref.shape, generator=g), dtype)
    ref.backward(dy)
    _, dw = ops.conv2d_backward(_nhwc(x, dtype, dev), w.detach().to(dev), _nhwc(dy, dtype, dev), stride=2, pad=3, need_dx=False)
    assert _rel(dw.cpu(), w.grad) < TOL[dtype]
