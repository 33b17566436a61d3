"""Shared helpers for the GPU parity tests (everything goes through the C ABI)."""
import ctypes as C

import numpy as np
import torch

from facenet_amd import _lib
from facenet_amd._lib import ConvDesc


# fixed-point accumulators of the C ABI (fn_acc_t, include/facenet_hip.h): value = integer * 2^-bits
ACC_STAT_BITS, ACC_GRAD_BITS = 20, 40


def to_acc(t, bits):
    """fp tensor -> int64 fixed-point accumulator contents (what a kernel's contributions would have summed to)."""
    return (t.double() * float(2 ** bits)).round().to(torch.int64)


def from_acc(t, bits):
    return (t.double() / float(2 ** bits)).float()


def stream():
    return torch.cuda.current_stream().cuda_stream


def ptr(t, off=0):
    return t.data_ptr() + off * t.element_size()


def lp_dtype(code):
    return torch.bfloat16 if code == _lib.FN_BF16 else torch.float16


def conv_desc(N, H, W, Cin, Cout, kh, kw, stride, ph, pw, dt, ld_x=None, ld_y=None):
    d = ConvDesc()
    d.N, d.H, d.W, d.Cin = N, H, W, Cin
    d.OH, d.OW, d.Cout = (H + 2 * ph - kh) // stride + 1, (W + 2 * pw - kw) // stride + 1, Cout
    d.KH, d.KW, d.stride, d.pad_h, d.pad_w = kh, kw, stride, ph, pw
    d.dtype = dt
    d.ld_x = ld_x or Cin
    d.ld_y = ld_y or Cout
    d.scale = 1.0
    return d


def ref_conv(x_nhwc, w_ohwi, stride, ph, pw):
    """fp32 reference on the CPU from the SAME low-precision-rounded operands."""
    x = x_nhwc.float().cpu().permute(0, 3, 1, 2)
    w = w_ohwi.float().cpu().permute(0, 3, 1, 2)
    return torch.nn.functional.conv2d(x, w, None, stride=stride, padding=(ph, pw)).permute(0, 2, 3, 1).contiguous()


def rel_err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


from tests.util_data import structured_images  # noqa: E402,F401
