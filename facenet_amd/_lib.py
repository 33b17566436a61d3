"""ctypes binding of libfacenet_hip.so (the C ABI in include/facenet_hip.h).

There is deliberately NO fallback: if the HIP library is missing, importing the
compute path raises.  ``import torch`` must come first so that the library
binds to the libamdhip64.so.7 torch already loaded (one HIP runtime per
process, shared streams).
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads the HIP runtime the library binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfacenet_hip.so")

FN_BF16, FN_F16 = 0, 1


class ConvDesc(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
        ("OH", C.c_int32), ("OW", C.c_int32), ("Cout", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad_h", C.c_int32), ("pad_w", C.c_int32),
        ("dtype", C.c_int32), ("ld_x", C.c_int32), ("ld_y", C.c_int32),
        ("relu", C.c_int32), ("accumulate", C.c_int32), ("out_f32", C.c_int32), ("ld_res", C.c_int32),
        ("scale", C.c_float), ("splits", C.c_int32), ("stats_sq_off", C.c_int32), ("stats_replicas", C.c_int32),
        ("stats_rep_stride", C.c_int32),
        ("x", C.c_void_p), ("w", C.c_void_p), ("y", C.c_void_p), ("dx", C.c_void_p), ("dw", C.c_void_p),
        ("bias", C.c_void_p), ("stats", C.c_void_p), ("resid", C.c_void_p),
        ("bn_y", C.c_void_p), ("bn_scale", C.c_void_p), ("bn_shift", C.c_void_p), ("bn_beta", C.c_void_p), ("bn_acc", C.c_void_p),
        ("ld_bn_y", C.c_int32), ("bn_sq_off", C.c_int32), ("bn_replicas", C.c_int32), ("bn_rep_stride", C.c_int32), ("bn_relu", C.c_int32),
        ("nrm_stats", C.c_void_p), ("nrm_beta", C.c_void_p),
        ("nrm_sq_off", C.c_int32), ("nrm_replicas", C.c_int32), ("nrm_rep_stride", C.c_int32), ("nrm_count", C.c_int32),
        ("nrm_eps", C.c_float), ("nrm_z", C.c_void_p), ("tile_fwd", C.c_int32), ("tile_dgrad", C.c_int32),
        ("dy2", C.c_void_p), ("w2", C.c_void_p), ("dy3", C.c_void_p), ("w3", C.c_void_p),
        ("Cout2", C.c_int32), ("ld_y2", C.c_int32), ("Cout3", C.c_int32), ("ld_y3", C.c_int32),
        ("rb_prev", C.c_void_p), ("rb_out", C.c_void_p), ("rb_dtrunk", C.c_void_p), ("rb_dup", C.c_void_p), ("rb_dbias", C.c_void_p),
        ("rb_scale", C.c_float), ("rb_accumulate", C.c_int32), ("prelu", C.c_void_p),
    ]


class FacenetHipError(RuntimeError):
    pass


_i, _f, _p, _l, _u = C.c_int, C.c_float, C.c_void_p, C.c_long, C.c_uint32

_SIGNATURES = {
    "fn_abi_version": [],
    "fn_conv2d_fwd": [C.POINTER(ConvDesc), _p],
    "fn_conv2d_dgrad": [C.POINTER(ConvDesc), _p],
    "fn_conv2d_wgrad": [C.POINTER(ConvDesc), _p],
    "fn_conv2d_variant": [C.POINTER(ConvDesc), _i],
    "fn_block35_infer": [_p, _p, _i, _p, _p, _p, _p, _p, _p, _f, _i, _i, _p],
    "fn_block35_infer_warm": [_p, _p, _i, _p, _p, _p, _p, _p, _p, _f, _i, _p, C.c_int64, _i, _p],
    "fn_block17_infer": [_p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _i, _i, _p],
    "fn_block17_infer_warm": [_p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _i, _p, C.c_int64, _i, _p],
    "fn_conv2d_arg_bytes": [],
    "fn_conv2d_group_build": [C.POINTER(ConvDesc), _i, _i, _i, _p, _p, _p],
    "fn_conv2d_grouped": [_p, _p, _i, _i, _i, _i, _i, _i, _p],
    "fn_conv2d_wgrad_arg_bytes": [],
    "fn_conv2d_wgrad_group_build": [C.POINTER(ConvDesc), _i, _i, _p, _p, _p, _p],
    "fn_conv2d_wgrad_reduce": [_p, _i, _p],
    "fn_conv2d_wgrad_grouped": [_p, _p, _i, _i, _i, _i, _p],
    "fn_image_normalize": [_p, _p, _p, _i, _i, _i, _i, _p],
    "fn_image_normalize_f32": [_p, _p, _p, _i, _i, _i, _i, _p],
    "fn_image_resize_bilinear": [_p, _i, _p, _i, _i, _i, _i, _i, _p],
    "fn_bn_finalize": [_p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _f, _f, _i, _p],
    "fn_crop_or_pad_u8": [_p, _p, _p, _p, _i, _i, _p],
    "fn_gather_images": [_p, _p, _p, _i, _i, _p],
    "fn_bn_relu_train_fwd": [_p, _i, _p, _i, _i, _i, _p, _i, _i, _i, _p, _p, _p, _p, _p, _f, _f, _i, _i, _p],
    "fn_bn_relu_train_bwd": [_p, _i, _p, _i, _i, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "fn_maxpool3x3s2_fwd": [_p, _i, _p, _i, _i, _i, _i, _i, _p, _i, _p],
    "fn_maxpool3x3s2_bwd": [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _p, _i, _i, _p],
    "fn_area_resize_crop": [_p, _i, _i, _p, _i, _i, _i, _i, _p, _i, _p],
    "fn_maxpool2d_fwd": [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "fn_area_resize_frame": [_p, _i, _i, _i, _i, _p, _p, _i, _p],
    "fn_mtcnn_candidates": [_p, C.c_long, _i, _f, _p, _p, _i, _i, _i, _p],
    "fn_nms_greedy_batch": [_p, _i, _p, _p, _p, _p, _i, _p, C.c_long, _p, _p, _p],
    "fn_nms_greedy": [_p, _i, _p, _i, C.c_double, _i, _p, C.c_long, _p, _p, _p],
    "fn_avgpool_fwd": [_p, _p, _i, _i, _i, _i, _p],
    "fn_avgpool_bwd": [_p, _p, _i, _i, _i, _i, _p],
    "fn_residual_bwd": [_p, _p, _p, _p, _p, _i, _i, _f, _i, _i, _i, _p],
    "fn_acc_to_float": [_p, _p, _l, _i, _p],
    "fn_head_bn_fwd": [_p, _p, _i, _i, _p, _p, _p, _p, _p, _i, _f, _f, _p],
    "fn_head_bn_bwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "fn_l2norm_fwd": [_p, _p, _i, _i, _f, _p],
    "fn_l2norm_bwd": [_p, _p, _p, _i, _i, _f, _p],
    "fn_cast_f32_to_lp": [_p, _p, _l, _i, _p],
    "fn_pairwise_sqdist": [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    "fn_select_triplets": [_p, _p, _i, _f, _i, _u, _i, _p, _p, _p],
    "fn_triplet_loss_fwd_bwd": [_p, _p, _p, _i, _i, _f, _p],
    "fn_confidence_counts": [_p, _p, _i, _i, _p, _i, _i, _p, _p, _p],
    "fn_softmax_xent_fwd_bwd": [_p, _i, _p, _p, _p, _i, _p, _i, _i, _f, _i, _p],
    "fn_adam_keras": [_p, _p, _p, _p, _p, _l, _l, _l, _p, _f, _f, _f, _f, _i, _p],
    "fn_adam_tick": [_p, _f, _f, _p],
    "fn_pack_transpose": [_p, _p, _p, _i, _i, _i, _p],
    "fn_fold_bn": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _i, _p],
}

EXPORTS = ["fn_last_error"] + list(_SIGNATURES)

_lib = None


def load():
    """Load (once) and return the ctypes handle; raise loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FacenetHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). facenet_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.fn_last_error.restype = C.c_char_p
    lib.fn_last_error.argtypes = []
    for name, args in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = args
    if lib.fn_abi_version() != 1:
        raise FacenetHipError("libfacenet_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().fn_last_error().decode("utf-8", "replace")
        # the reference raises ValueError for bad modes / metrics (facenet.py:82, statistics.py:55)
        if rc == -1:
            raise ValueError(msg or what)
        raise FacenetHipError(f"{what}: rc={rc}: {msg}")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.bfloat16:
        return FN_BF16
    if dt == torch.float16:
        return FN_F16
    raise ValueError(f"unsupported low-precision dtype {dt}")
