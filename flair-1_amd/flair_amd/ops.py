"""Operator-level wrappers over the C ABI (used by the parity tests and by host code).

Activations are NHWC tensors in the compute dtype (torch.float32 / torch.bfloat16); weights are
PyTorch OIHW fp32.  Every call goes through libflair_hip.so; nothing here computes on the host.
"""
from __future__ import annotations

import torch

from . import _lib as L


def _ws(nbytes, device):
    return torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)


def _dt(t):
    return L.dtype_code(t.dtype)


def nchw_to_nhwc(x: torch.Tensor, dtype, cpad=None) -> torch.Tensor:
    N, C, H, W = x.shape
    dt = L.dtype_code(dtype)
    cpad = cpad or ((C + 7) // 8 * 8)
    y = torch.empty(N, H, W, cpad, dtype=L.torch_dtype(dt), device=x.device)
    L.check(L.lib().flair_nchw_to_nhwc(dt, L.ptr(x.float().contiguous()), L.ptr(y), N, C, H, W, cpad, L.stream()), "nchw_to_nhwc")
    return y


def nhwc_to_nchw(x: torch.Tensor, C=None) -> torch.Tensor:
    N, H, W, Cp = x.shape
    C = C or Cp
    y = torch.empty(N, C, H, W, dtype=torch.float32, device=x.device)
    L.check(L.lib().flair_nhwc_to_nchw(_dt(x), L.ptr(x), L.ptr(y), N, C, H, W, Cp, L.stream()), "nhwc_to_nchw")
    return y


def conv2d_forward(x0, w, bias=None, stride=1, pad=1, x1=None, up0=False, want_nchw=False, want_stats=False):
    """y = conv2d(cat([up2(x0) if up0 else x0, x1], C), w) on NHWC inputs; returns (y_nhwc, y_nchw, stats)."""
    N, H, W, C0 = x0.shape
    C1 = x1.shape[3] if x1 is not None else 0
    Cout, Cin, R, _ = w.shape
    assert Cin == C0 + C1
    dt = _dt(x0)
    Hin, Win = (2 * H, 2 * W) if up0 else (H, W)
    Ho, Wo = (Hin + 2 * pad - R) // stride + 1, (Win + 2 * pad - R) // stride + 1
    l = L.lib()
    ws = _ws(l.flair_conv2d_workspace_bytes(dt, N, H, W, C0, C1, int(up0), Cout, R, stride, pad), x0.device)
    y = torch.empty(N, Ho, Wo, Cout, dtype=x0.dtype, device=x0.device) if Cout % 8 == 0 else None
    yn = torch.empty(N, Cout, Ho, Wo, dtype=torch.float32, device=x0.device) if (want_nchw or y is None) else None
    st = torch.empty(2, Cout, dtype=torch.float32, device=x0.device) if want_stats else None
    L.check(l.flair_conv2d_forward(dt, L.ptr(x0), L.ptr(x1), N, H, W, C0, C1, int(up0), L.ptr(w), L.ptr(bias), Cout, R,
                                   stride, pad, L.ptr(y), L.ptr(yn), L.ptr(st), L.ptr(ws), ws.numel(), L.stream()),
            "conv2d_forward")
    return y, yn, st


def conv2d_backward(x, w, dy, stride=1, pad=1, need_dx=True, need_dw=True):
    N, H, W, Cin = x.shape
    Cout, _, R, _ = w.shape
    dt = _dt(x)
    l = L.lib()
    ws = _ws(l.flair_conv2d_workspace_bytes(dt, N, H, W, Cin, 0, 0, Cout, R, stride, pad), x.device)
    dx = torch.empty_like(x) if need_dx else None
    dw = torch.empty_like(w) if need_dw else None
    L.check(l.flair_conv2d_backward(dt, L.ptr(x), N, H, W, Cin, L.ptr(w), Cout, R, stride, pad, L.ptr(dy), L.ptr(dx),
                                    L.ptr(dw), L.ptr(ws), ws.numel(), L.stream()), "conv2d_backward")
    return dx, dw


def bn_relu_forward(y, gamma, beta, running_mean, running_var, training=True, residual=None, relu=True):
    C = y.shape[-1]
    rows = y.numel() // C
    out = torch.empty_like(y)
    mean = torch.empty(C, dtype=torch.float32, device=y.device)
    invstd = torch.empty(C, dtype=torch.float32, device=y.device)
    ws = _ws(1024 * 2 * C * 4 + 16 * C * 4 + 8192, y.device)
    L.check(L.lib().flair_bn_relu_forward(_dt(y), L.ptr(y), rows, C, L.ptr(gamma), L.ptr(beta), L.ptr(running_mean),
                                          L.ptr(running_var), int(training), L.ptr(residual), int(relu), L.ptr(out),
                                          L.ptr(mean), L.ptr(invstd), L.ptr(ws), ws.numel(), L.stream()), "bn_relu_forward")
    return out, mean, invstd


def bn_relu_backward(dout, out, y, gamma, mean, invstd, relu=True, want_dres=False):
    C = y.shape[-1]
    rows = y.numel() // C
    dy = torch.empty_like(y)
    dres = torch.empty_like(y) if want_dres else None
    dgamma = torch.empty(C, dtype=torch.float32, device=y.device)
    dbeta = torch.empty(C, dtype=torch.float32, device=y.device)
    ws = _ws(1024 * 2 * C * 4 + 16 * C * 4 + 8192, y.device)
    L.check(L.lib().flair_bn_relu_backward(_dt(y), L.ptr(dout), L.ptr(out), L.ptr(y), rows, C, L.ptr(gamma), L.ptr(mean),
                                           L.ptr(invstd), int(relu), L.ptr(dy), L.ptr(dres), L.ptr(dgamma), L.ptr(dbeta),
                                           L.ptr(ws), ws.numel(), L.stream()), "bn_relu_backward")
    return dy, dres, dgamma, dbeta


def maxpool_forward(x):
    N, H, W, C = x.shape
    y = torch.empty(N, H // 2, W // 2, C, dtype=x.dtype, device=x.device)
    idx = torch.empty(N, H // 2, W // 2, C, dtype=torch.uint8, device=x.device)
    L.check(L.lib().flair_maxpool_forward(_dt(x), L.ptr(x), L.ptr(y), L.ptr(idx), N, H, W, C, L.stream()), "maxpool_forward")
    return y, idx


def maxpool_backward(dy, idx, H, W):
    N, _, _, C = dy.shape
    dx = torch.empty(N, H, W, C, dtype=dy.dtype, device=dy.device)
    L.check(L.lib().flair_maxpool_backward(_dt(dy), L.ptr(dy), L.ptr(idx), L.ptr(dx), N, H, W, C, L.stream()), "maxpool_backward")
    return dx


_LABEL_KIND = {torch.uint8: 0, torch.int32: 1, torch.int64: 2}


def ce_head(logits, labels, weight=None, want_dlogits=True, want_preds="u8", confmat=None, want_targets=False,
            dlogits_nhwc=None, dlogits_ld=0):
    """Fused CE + argmax(softmax) + confusion matrix.  labels: (B,H,W) uint8/int32/int64 or fp32 one-hot (B,C,H,W)."""
    B, Cc, H, W = logits.shape
    kind = 3 if labels.dtype == torch.float32 else _LABEL_KIND[labels.dtype]
    dev = logits.device
    loss = torch.empty((), dtype=torch.float32, device=dev)
    dl = torch.empty_like(logits) if want_dlogits else None
    pu8 = torch.empty(B, H, W, dtype=torch.uint8, device=dev) if want_preds == "u8" else None
    pi64 = torch.empty(B, H, W, dtype=torch.int64, device=dev) if want_preds == "i64" else None
    tg = torch.empty(B, H, W, dtype=torch.int32, device=dev) if want_targets else None
    l = L.lib()
    ws = _ws(l.flair_ce_workspace_bytes(B, H, W), dev)
    L.check(l.flair_ce_head(L.ptr(logits), L.ptr(labels), kind, L.ptr(weight), B, Cc, H, W, L.ptr(loss), L.ptr(dl),
                            L.ptr(dlogits_nhwc), _dt(dlogits_nhwc) if dlogits_nhwc is not None else 0, dlogits_ld,
                            L.ptr(pu8), L.ptr(pi64), L.ptr(tg), L.ptr(confmat), L.ptr(ws), L.stream()), "ce_head")
    return loss, dl, (pu8 if pu8 is not None else pi64), tg


def softmax_argmax(logits, want="i64", want_maxprob=False):
    B, Cc, H, W = logits.shape
    dev = logits.device
    pu8 = torch.empty(B, H, W, dtype=torch.uint8, device=dev) if want == "u8" else None
    pi64 = torch.empty(B, H, W, dtype=torch.int64, device=dev) if want == "i64" else None
    mp = torch.empty(B, H, W, dtype=torch.float32, device=dev) if want_maxprob else None
    L.check(L.lib().flair_softmax_argmax(L.ptr(logits), B, Cc, H, W, L.ptr(pu8), L.ptr(pi64), L.ptr(mp), L.stream()), "softmax_argmax")
    p = pu8 if pu8 is not None else pi64
    return (p, mp) if want_maxprob else p


def confmat_update(confmat, target, pred):
    Cc = confmat.shape[0]
    L.check(L.lib().flair_confmat_update(L.ptr(target), _LABEL_KIND[target.dtype], L.ptr(pred), _LABEL_KIND[pred.dtype],
                                         target.numel(), Cc, L.ptr(confmat), L.stream()), "confmat_update")
    return confmat


def jaccard(confmat):
    Cc = confmat.shape[0]
    dev = confmat.device
    per = torch.empty(Cc, dtype=torch.float32, device=dev)
    w = torch.empty((), dtype=torch.float32, device=dev)
    m = torch.empty((), dtype=torch.float32, device=dev)
    L.check(L.lib().flair_jaccard(L.ptr(confmat), Cc, L.ptr(per), L.ptr(w), L.ptr(m), L.stream()), "jaccard")
    return per, w, m


def sgd_step_(params_flat, grads_flat, lr):
    L.check(L.lib().flair_sgd_step(L.ptr(params_flat), L.ptr(grads_flat), params_flat.numel(), float(lr), L.stream()), "sgd_step")
