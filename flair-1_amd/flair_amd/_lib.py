"""ctypes binding of libflair_hip.so (C ABI declared in include/flair_hip.h).

There is NO CPU fallback: if the shared library is missing or a tensor is not on a HIP device the
product path raises.  (Build: ``python flair-1_amd/build.py`` or ``__graft_entry__.build()``.)
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FLAIR_HIP_LIB") or os.path.join(_HERE, "libflair_hip.so")   # override: A/B of two builds on one box

_lib = None

vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes); must list every symbol of include/flair_hip.h
PROTOTYPES = {
    "flair_strerror": (C.c_char_p, [i32]),
    "flair_version": (i32, []),
    "flair_unet_create": (i32, [C.POINTER(vp), i32, i32, i32]),
    "flair_unet_destroy": (None, [vp]),
    "flair_unet_param_count": (i64, [vp]),
    "flair_unet_buffer_count": (i64, [vp]),
    "flair_unet_num_tensors": (i32, [vp]),
    "flair_unet_tensor_info": (i32, [vp, i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(i32), C.POINTER(i64),
                                     C.POINTER(i32), C.POINTER(i32)]),
    "flair_unet_stage_range": (i32, [vp, i32, C.POINTER(i64), C.POINTER(i64)]),
    "flair_unet_workspace_bytes": (i64, [vp, i32, i32, i32, i32]),
    "flair_unet_forward": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp]),
    "flair_unet_backward": (i32, [vp, vp, vp, vp, vp, vp, sz, vp, C.POINTER(vp)]),
    "flair_unet_head_ld": (i32, [vp]),
    "flair_unet_encoder_forward": (i32, [vp, vp, vp, vp, C.POINTER(vp), i32, i32, i32, i32, vp, sz, vp]),
    "flair_unet_decoder_forward": (i32, [vp, vp, vp, C.POINTER(vp), vp, i32, i32, i32, i32, vp, sz, vp]),
    "flair_unet_head_forward": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp]),
    "flair_unet_head_backward": (i32, [vp, vp, vp, vp, vp, vp, sz, vp]),
    "flair_unet_decoder_backward": (i32, [vp, vp, vp, C.POINTER(vp), vp, vp, sz, vp]),
    "flair_unet_encoder_backward": (i32, [vp, vp, C.POINTER(vp), vp, vp, sz, vp]),
    "flair_ce_workspace_bytes": (sz, [i32, i32, i32]),
    "flair_ce_head": (i32, [vp, vp, i32, vp, i32, i32, i32, i32, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp]),
    "flair_unet_logits_nhwc": (vp, [vp]),
    "flair_ce_head_nhwc": (i32, [vp, i32, i32, vp, i32, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
    "flair_softmax_argmax_nhwc": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
    "flair_softmax_argmax": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, vp]),
    "flair_confmat_update": (i32, [vp, i32, vp, i32, i64, i32, vp, vp]),
    "flair_jaccard": (i32, [vp, i32, vp, vp, vp, vp]),
    "flair_confmat_masks": (i32, [vp, vp, i64, i32, i32, vp, vp]),
    "flair_feed_tiles": (i32, [vp, vp, vp, i32, i32, i32, i32, C.POINTER(i32), i32, i32, C.POINTER(C.c_double),
                               C.POINTER(C.c_double), i32, vp, vp, vp]),
    "flair_detect_convert": (i32, [vp, i32, i32, i32, i32, i32, vp, vp]),
    "flair_gather_tiles": (i32, [vp, i32, i32, i32, vp, i32, i32, C.POINTER(i32), i32, i32, C.POINTER(C.c_double),
                                 C.POINTER(C.c_double), vp, vp]),
    "flair_detect_stitch": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp]),
    "flair_sgd_step": (i32, [vp, vp, i64, f32, vp]),
    "flair_add_rowvec_nchw": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "flair_conv2d_workspace_bytes": (sz, [i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32]),
    "flair_conv2d_forward": (i32, [i32, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, i32, i32, i32, vp, vp, vp,
                                   vp, sz, vp]),
    "flair_conv2d_backward": (i32, [i32, vp, i32, i32, i32, i32, vp, i32, i32, i32, i32, vp, vp, vp, vp, sz, vp]),
    "flair_bn_relu_forward": (i32, [i32, vp, i64, i32, vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, sz, vp]),
    "flair_bn_relu_backward": (i32, [i32, vp, vp, vp, i64, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, sz, vp]),
    "flair_maxpool_forward": (i32, [i32, vp, vp, vp, i32, i32, i32, i32, vp]),
    "flair_maxpool_backward": (i32, [i32, vp, vp, vp, i32, i32, i32, i32, vp]),
    "flair_nchw_to_nhwc": (i32, [i32, vp, vp, i32, i32, i32, i32, i32, vp]),
    "flair_nhwc_to_nchw": (i32, [i32, vp, vp, i32, i32, i32, i32, i32, vp]),
    "flair_profile_start": (i32, [i32]),
    "flair_profile_stop": (i32, []),
    "flair_profile_kernel": (i32, [i32, C.c_char_p, i32, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double),
                                   C.POINTER(C.c_double)]),
    "flair_metadata_mlp_forward": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "flair_metadata_mlp_backward": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp]),
    "flair_unet_reuse_constants": (i32, [vp, i32]),
    "flair_unet_want_preds": (i32, [vp, vp, vp]),
    "flair_detect_stitch_preds": (i32, [vp, vp, i32, i32, i32, vp, vp, i32, i32, vp]),
    "flair_segformer_create": (i32, [C.POINTER(vp), i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), i32, i32]),
    "flair_segformer_destroy": (None, [vp]),
    "flair_segformer_param_count": (i64, [vp]),
    "flair_segformer_num_tensors": (i32, [vp]),
    "flair_segformer_tensor_info": (i32, [vp, i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(i32), C.POINTER(i64), C.POINTER(i32)]),
    "flair_segformer_workspace_bytes": (i64, [vp, i32, i32, i32]),
    "flair_segformer_weights_changed": (None, [vp]),
    "flair_segformer_forward": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp, sz, vp]),
    "flair_tune_set": (i32, [C.c_char_p, i32]),
    "flair_debug_buffer": (i32, [vp]),
}


class FlairHipError(RuntimeError):
    pass


def lib():
    """Load the HIP library once; fail loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FlairHipError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
                "Build it with `python flair-1_amd/build.py`.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().flair_strerror(rc).decode()
        raise FlairHipError(f"{what}: {msg} (code {rc})" if what else f"{msg} (code {rc})")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses host tensors: the product path is HIP-only."""
    if t is None:
        return None
    if not t.is_cuda:
        raise FlairHipError("flair_amd kernels need tensors on a HIP device (no CPU fallback)")
    if not t.is_contiguous():
        raise FlairHipError("flair_amd kernels need contiguous tensors")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


DT_F32, DT_BF16 = 0, 1


def torch_dtype(dt: int):
    return torch.float32 if dt == DT_F32 else torch.bfloat16


def dtype_code(name) -> int:
    if name in (0, "f32", "fp32", "float32", torch.float32):
        return DT_F32
    if name in (1, "bf16", "bfloat16", torch.bfloat16):
        return DT_BF16
    raise ValueError(f"unsupported compute dtype {name!r} (use 'f32' or 'bf16')")
