"""Implicit-GEMM convolution kernels (fwd / dgrad / wgrad) against an fp32 CPU convolution on the same
rounded operands.  Shapes cover every kernel size / stride / padding of Inception-ResNet-v1, ragged M,
odd channel counts (80, 10575-like), channel slices (ld > C) and the fused epilogues."""
import ctypes as C

import numpy as np
import pytest
import torch

from facenet_amd import _lib
from tests.util import ACC_GRAD_BITS, ACC_STAT_BITS, conv_desc, from_acc, lp_dtype, ptr, ref_conv, rel_err, stream, to_acc

pytestmark = pytest.mark.gpu

CASES = [
    # N, H, W, Cin, Cout, kh, kw, stride, ph, pw
    (2, 19, 19, 8, 32, 3, 3, 2, 0, 0),     # stem 1a-like (Cin padded 3->8)
    (2, 15, 15, 32, 64, 3, 3, 1, 0, 0),    # 2b
    (3, 9, 9, 64, 80, 1, 1, 1, 0, 0),      # 3b (Cout=80)
    (2, 11, 11, 80, 192, 3, 3, 1, 0, 0),   # 4a (Cin=80: K tiles straddle taps)
    (2, 17, 17, 256, 32, 1, 1, 1, 0, 0),   # block35 1x1
    (2, 17, 17, 32, 32, 3, 3, 1, 1, 1),    # block35 3x3 same
    (2, 17, 17, 192, 256, 3, 3, 2, 0, 0),  # reduction stride-2
    (3, 8, 8, 128, 128, 1, 7, 1, 0, 3),    # block17 1x7
    (3, 8, 8, 128, 128, 7, 1, 1, 3, 0),    # block17 7x1
    (5, 3, 3, 192, 192, 1, 3, 1, 0, 1),    # block8 1x3
    (5, 3, 3, 192, 192, 3, 1, 1, 1, 0),    # block8 3x1
    (5, 3, 3, 384, 1792, 1, 1, 1, 0, 0),   # block8 up
    (7, 1, 1, 1792, 128, 1, 1, 1, 0, 0),   # Dense 1792 -> E
    # halo-tile kernel (stride-1 3x3 on maps >= 30x30): ragged 8x16 tiles, partial 32-channel slices, both column-tile widths
    (2, 37, 37, 80, 192, 3, 3, 1, 0, 0),   # 4a at full map size (Cin=80: the third slice is half empty; dgrad: Cout tile 32 x 3)
    (2, 40, 33, 32, 64, 3, 3, 1, 1, 1),    # 'same' padding: zero fill on all four sides
    (1, 45, 39, 32, 32, 3, 3, 1, 0, 0),    # 2a-like: 32-wide column tile
    (2, 35, 35, 192, 80, 3, 3, 1, 0, 0),   # Cout=80 (three 32-wide column tiles, the last half empty), six input slices
]
HALO_CASES = CASES[-4:]
HALO = 9000000      # fn_conv_desc.tile_*: ask for the halo-tile kernel (production picks it only for <= 64 source channels, where it wins)


def _halo(d, case):
    """The halo cases run on conv_halo_kernel whatever the production heuristic would choose (per-descriptor request: the model,
    golden and training tests dispatch exactly like bench.py and users do)."""
    if case in HALO_CASES:
        d.tile_fwd = d.tile_dgrad = HALO
    return d


def test_halo_cases_dispatch_to_the_halo_kernel(lib):
    """The cases above are there to exercise conv_halo_kernel: fn_conv2d_variant reports 9000000 + BN for them (forward and
    data gradient), and an explicit tile sends the same layer to the implicit-GEMM kernel."""
    for (N, H, W, Cin, Cout, kh, kw, s, ph, pw) in HALO_CASES:
        d = conv_desc(N, H, W, Cin, Cout, kh, kw, s, ph, pw, _lib.FN_BF16)
        t = torch.zeros(16, device="cuda")
        d.x = d.w = d.y = d.dx = ptr(t)
        prod = Cin <= 64, Cout <= 64       # the production heuristic: at most 64 SOURCE channels (forward: Cin, data gradient: Cout)
        assert (lib.fn_conv2d_variant(C.byref(d), 0) >= 9000000) == prod[0] and (lib.fn_conv2d_variant(C.byref(d), 1) >= 9000000) == prod[1]
        d.tile_fwd = d.tile_dgrad = HALO
        assert lib.fn_conv2d_variant(C.byref(d), 0) >= 9000000 and lib.fn_conv2d_variant(C.byref(d), 1) >= 9000000
        d.tile_fwd = d.tile_dgrad = 64064
        assert lib.fn_conv2d_variant(C.byref(d), 0) % 1000000 == 64064 and lib.fn_conv2d_variant(C.byref(d), 1) % 1000000 == 64064   # (+ 2e6: in-launch split-K)
    d = conv_desc(2, 17, 17, 32, 32, 3, 3, 1, 1, 1, _lib.FN_BF16)          # small map: implicit GEMM
    d.x = d.w = d.y = d.dx = ptr(torch.zeros(16, device="cuda"))
    assert lib.fn_conv2d_variant(C.byref(d), 0) < 9000000
    d.tile_fwd = HALO                                                       # an explicit request is honoured on any map size ...
    assert lib.fn_conv2d_variant(C.byref(d), 0) >= 9000000
    d = conv_desc(2, 17, 17, 64, 64, 1, 1, 1, 0, 0, _lib.FN_BF16)           # ... but not for a layer the kernel cannot run
    d.x = d.w = d.y = d.dx = ptr(torch.zeros(16, device="cuda"))
    d.tile_fwd = HALO
    assert lib.fn_conv2d_variant(C.byref(d), 0) < 0
    with pytest.raises(_lib.FacenetHipError):
        _lib.check(lib.fn_conv2d_fwd(C.byref(d), stream()))


def _mk(shape, dt, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(lp_dtype(dt)).cuda()


@pytest.mark.parametrize("dt", [_lib.FN_BF16, _lib.FN_F16])
@pytest.mark.parametrize("case", CASES)
def test_conv_fwd(lib, case, dt):
    N, H, W, Cin, Cout, kh, kw, s, ph, pw = case
    x = _mk((N, H, W, Cin), dt, seed=1)
    w = _mk((Cout, kh, kw, Cin), dt, 0.1, seed=2)
    d = _halo(conv_desc(N, H, W, Cin, Cout, kh, kw, s, ph, pw, dt), case)
    y = torch.full((N, d.OH, d.OW, Cout), 7.0, dtype=lp_dtype(dt), device="cuda")
    reps = 4
    stats_r = torch.zeros(reps, 2 * Cout, dtype=torch.int64, device="cuda")      # fixed-point accumulators (fn_acc_t)
    d.x, d.w, d.y, d.stats, d.stats_sq_off, d.stats_replicas, d.stats_rep_stride = ptr(x), ptr(w), ptr(y), ptr(stats_r), Cout, reps, 2 * Cout
    _lib.check(lib.fn_conv2d_fwd(C.byref(d), stream()))
    torch.cuda.synchronize()
    stats = from_acc(stats_r.sum(0), ACC_STAT_BITS)            # row tiles are spread over the accumulator replicas
    first = stats_r.clone()
    for _ in range(2):                 # integer accumulation: the statistics have the same bits every run, in every replica
        stats_r.zero_()
        _lib.check(lib.fn_conv2d_fwd(C.byref(d), stream()))
        torch.cuda.synchronize()
        assert torch.equal(stats_r, first)
    ref = ref_conv(x, w, s, ph, pw)
    assert rel_err(y, ref) < (6e-3 if dt == _lib.FN_BF16 else 8e-4)
    # BatchNorm statistics come from the fp32 accumulators
    M = N * d.OH * d.OW
    s_ref = ref.reshape(M, Cout).sum(0)
    q_ref = (ref.reshape(M, Cout) ** 2).sum(0)
    assert torch.allclose(stats[:Cout].cpu(), s_ref, rtol=2e-3, atol=2e-3 * float(s_ref.abs().max()))
    assert torch.allclose(stats[Cout:].cpu(), q_ref, rtol=2e-3, atol=1e-3)


@pytest.mark.parametrize("dt", [_lib.FN_BF16])
def test_conv_fwd_epilogues_and_slices(lib, dt):
    """bias + residual scale-add + ReLU, output into a channel slice, input from a channel slice, fp32 output."""
    N, H, W, Cin, Cout = 2, 8, 8, 64, 96
    xb = _mk((N, H, W, 160), dt, seed=3)           # input slice [32:96] of a 160-channel buffer
    w = _mk((Cout, 1, 1, Cin), dt, 0.2, seed=4)
    bias = torch.randn(Cout).cuda()
    res = _mk((N, H, W, Cout), dt, seed=5)
    yb = torch.zeros(N, H, W, 256, dtype=lp_dtype(dt), device="cuda")   # output slice [128:224]
    d = conv_desc(N, H, W, Cin, Cout, 1, 1, 1, 0, 0, dt, ld_x=160, ld_y=256)
    d.x, d.w, d.y, d.bias, d.resid, d.ld_res, d.scale, d.relu = ptr(xb, 32), ptr(w), ptr(yb, 128), ptr(bias), ptr(res), Cout, 0.17, 1
    _lib.check(lib.fn_conv2d_fwd(C.byref(d), stream()))
    torch.cuda.synchronize()
    ref = torch.relu(res.float().cpu() + 0.17 * (ref_conv(xb[..., 32:96], w, 1, 0, 0) + bias.cpu()))
    assert rel_err(yb[..., 128:224], ref) < 6e-3
    assert float(yb[..., :128].abs().max()) == 0 and float(yb[..., 224:].abs().max()) == 0   # neighbours untouched
    # fp32 output with a ragged Cout (classifier-like: 203 of 208 columns)
    Cr, Cp = 203, 208
    w2 = _mk((Cp, 1, 1, Cin), dt, 0.2, seed=6)
    w2[Cr:] = 0
    x2 = _mk((5, 1, 1, Cin), dt, seed=7)
    y2 = torch.zeros(5, Cp, dtype=torch.float32, device="cuda")
    d2 = conv_desc(5, 1, 1, Cin, Cp, 1, 1, 1, 0, 0, dt)
    d2.x, d2.w, d2.y, d2.out_f32 = ptr(x2), ptr(w2), ptr(y2), 1
    _lib.check(lib.fn_conv2d_fwd(C.byref(d2), stream()))
    torch.cuda.synchronize()
    assert rel_err(y2.view(5, 1, 1, Cp), ref_conv(x2, w2, 1, 0, 0)) < 1e-5


@pytest.mark.parametrize("dt", [_lib.FN_BF16, _lib.FN_F16])
@pytest.mark.parametrize("case", CASES[1:])
def test_conv_dgrad(lib, case, dt):
    N, H, W, Cin, Cout, kh, kw, s, ph, pw = case
    d = _halo(conv_desc(N, H, W, Cin, Cout, kh, kw, s, ph, pw, dt), case)
    dy = _mk((N, d.OH, d.OW, Cout), dt, seed=11)
    w = _mk((Cout, kh, kw, Cin), dt, 0.1, seed=12)
    wt = torch.zeros_like(w).view(-1)
    table = torch.tensor([[0, Cout, kh * kw * Cin, kh * kw, Cin, -1, -1, 0]], dtype=torch.int32, device="cuda")
    _lib.check(lib.fn_pack_transpose(ptr(w), ptr(wt), ptr(table), 1, w.numel(), dt, stream()))
    dx = torch.full((N, H, W, Cin), 3.0, dtype=lp_dtype(dt), device="cuda")
    d.y, d.w, d.dx = ptr(dy), ptr(wt), ptr(dx)
    _lib.check(lib.fn_conv2d_dgrad(C.byref(d), stream()))
    torch.cuda.synchronize()
    assert torch.equal(wt.view(Cin, kh * kw, Cout).cpu(), w.view(Cout, kh * kw, Cin).permute(2, 1, 0).cpu())
    xr = torch.zeros(N, Cin, H, W, requires_grad=True)
    yr = torch.nn.functional.conv2d(xr, w.float().cpu().permute(0, 3, 1, 2), None, stride=s, padding=(ph, pw))
    yr.backward(dy.float().cpu().permute(0, 3, 1, 2))
    ref = xr.grad.permute(0, 2, 3, 1)
    assert rel_err(dx, ref) < (6e-3 if dt == _lib.FN_BF16 else 8e-4)
    # accumulate mode adds on top
    d.accumulate = 1
    _lib.check(lib.fn_conv2d_dgrad(C.byref(d), stream()))
    torch.cuda.synchronize()
    assert rel_err(dx, 2 * ref) < (1e-2 if dt == _lib.FN_BF16 else 2e-3)


@pytest.mark.parametrize("dt", [_lib.FN_BF16, _lib.FN_F16])
@pytest.mark.parametrize("case", CASES)
def test_conv_wgrad(lib, case, dt):
    N, H, W, Cin, Cout, kh, kw, s, ph, pw = case
    d = conv_desc(N, H, W, Cin, Cout, kh, kw, s, ph, pw, dt)
    x = _mk((N, H, W, Cin), dt, seed=21)
    dy = _mk((N, d.OH, d.OW, Cout), dt, seed=22)
    dw = torch.zeros(Cout, kh, kw, Cin, dtype=torch.float32, device="cuda")
    d.x, d.y, d.dw = ptr(x), ptr(dy), ptr(dw)
    _lib.check(lib.fn_conv2d_wgrad(C.byref(d), stream()))
    torch.cuda.synchronize()
    wr = torch.zeros(Cout, Cin, kh, kw, requires_grad=True)
    yr = torch.nn.functional.conv2d(x.float().cpu().permute(0, 3, 1, 2), wr, None, stride=s, padding=(ph, pw))
    yr.backward(dy.float().cpu().permute(0, 3, 1, 2))
    ref = wr.grad.permute(0, 2, 3, 1)
    assert rel_err(dw, ref) < 2e-5     # operands are exact in low precision, accumulation is fp32
    # forced split-K
    dw.zero_()
    d.splits = 3
    _lib.check(lib.fn_conv2d_wgrad(C.byref(d), stream()))
    torch.cuda.synchronize()
    assert rel_err(dw, ref) < 2e-5


TILE_CASES = [(2, 37, 37, 80, 192, 3, 3, 1, 0, 0),      # 4a: general gather, K tiles straddle taps, 192 = 1.5 x 128 columns
              (2, 35, 35, 192, 256, 3, 3, 2, 0, 0),     # 4b: stride 2 (parity-class data gradient)
              (3, 17, 17, 256, 160, 1, 1, 1, 0, 0),     # 1x1 fast path, ragged rows (867) and a ragged column tile
              (2, 17, 17, 192, 192, 3, 3, 1, 1, 1)]     # 'same' padding


@pytest.mark.parametrize("tile", [128128, 128064, 64128, 32032])
@pytest.mark.parametrize("case", TILE_CASES)
def test_conv_explicit_tiles(lib, case, tile):
    """Every tile a caller may pin (fn_conv_desc.tile_fwd / tile_dgrad) gives the convolution, its BatchNorm statistics and the
    data gradient: forward and data gradient against fp32 references on the same rounded operands."""
    dt = _lib.FN_BF16
    N, H, W, Cin, Cout, kh, kw, s, ph, pw = case
    x = _mk((N, H, W, Cin), dt, seed=71)
    w = _mk((Cout, kh, kw, Cin), dt, 0.1, seed=72)
    d = conv_desc(N, H, W, Cin, Cout, kh, kw, s, ph, pw, dt)
    d.tile_fwd = d.tile_dgrad = tile
    assert lib.fn_conv2d_variant(C.byref(conv_probe(d)), 0) % 1000000 == tile
    y = torch.zeros(N, d.OH, d.OW, Cout, dtype=lp_dtype(dt), device="cuda")
    reps = 4
    stats_r = torch.zeros(reps, 2 * Cout, dtype=torch.int64, device="cuda")
    d.x, d.w, d.y, d.stats, d.stats_sq_off, d.stats_replicas, d.stats_rep_stride = ptr(x), ptr(w), ptr(y), ptr(stats_r), Cout, reps, 2 * Cout
    _lib.check(lib.fn_conv2d_fwd(C.byref(d), stream()))
    torch.cuda.synchronize()
    ref = ref_conv(x, w, s, ph, pw)
    assert rel_err(y, ref) < 6e-3
    M = N * d.OH * d.OW
    stats = from_acc(stats_r.sum(0), ACC_STAT_BITS).cpu()
    assert torch.allclose(stats[:Cout], ref.reshape(M, Cout).sum(0), rtol=2e-3, atol=2e-3 * float(ref.reshape(M, Cout).sum(0).abs().max()))
    assert torch.allclose(stats[Cout:], (ref.reshape(M, Cout) ** 2).sum(0), rtol=2e-3, atol=1e-3)
    # data gradient
    dy = _mk((N, d.OH, d.OW, Cout), dt, seed=73)
    wt = torch.zeros_like(w).view(-1)
    table = torch.tensor([[0, Cout, kh * kw * Cin, kh * kw, Cin, -1, -1, 0]], dtype=torch.int32, device="cuda")
    _lib.check(lib.fn_pack_transpose(ptr(w), ptr(wt), ptr(table), 1, w.numel(), dt, stream()))
    dx = torch.full((N, H, W, Cin), 3.0, dtype=lp_dtype(dt), device="cuda")
    g = conv_desc(N, H, W, Cin, Cout, kh, kw, s, ph, pw, dt)
    g.tile_dgrad = tile
    g.y, g.w, g.dx = ptr(dy), ptr(wt), ptr(dx)
    _lib.check(lib.fn_conv2d_dgrad(C.byref(g), stream()))
    torch.cuda.synchronize()
    xr = torch.zeros(N, Cin, H, W, requires_grad=True)
    yr = torch.nn.functional.conv2d(xr, w.float().cpu().permute(0, 3, 1, 2), None, stride=s, padding=(ph, pw))
    yr.backward(dy.float().cpu().permute(0, 3, 1, 2))
    assert rel_err(dx, xr.grad.permute(0, 2, 3, 1)) < 6e-3


def conv_probe(d):
    """A copy of the descriptor with dummy pointers where fn_conv2d_variant needs them set."""
    e = _lib.ConvDesc.from_buffer_copy(d)
    e.x = e.w = e.y = e.dx = 4096
    return e


TAPS_CASES = [
    # N, H, W, Cin, Cout, kh, kw, stride, ph, pw, ld_x, ld_y, variant
    (2, 19, 19, 32, 64, 3, 3, 1, 0, 0, 32, 64, 5064090),       # 2b-like, 'valid'
    (2, 17, 17, 32, 32, 3, 3, 1, 1, 1, 96, 32, 5064090),       # block35 3x3 'same', 32 couts, x is a channel slice of a wider buffer
    (2, 37, 37, 80, 192, 3, 3, 1, 0, 0, 80, 192, 5064090),     # 4a: Cin = 80 -> the third 32-channel slice is half empty
    (2, 35, 35, 192, 256, 3, 3, 2, 0, 0, 192, 640, 5064090),   # 4b-like stride 2, dy is a channel slice of a concat buffer
    (2, 17, 17, 256, 384, 3, 3, 2, 0, 0, 256, 384, 5064090),   # reduction_a stride 2 (17 -> 8)
    (3, 8, 8, 128, 128, 1, 7, 1, 0, 3, 128, 128, 5064090),     # block17 1x7
    (3, 8, 8, 128, 128, 7, 1, 1, 3, 0, 128, 256, 5064090),     # block17 7x1
    (1, 45, 39, 32, 40, 3, 3, 1, 1, 1, 32, 40, 5064090),       # ragged tiles, Cout = 40 (one partly empty cout tile)
]


@pytest.mark.parametrize("dt", [_lib.FN_BF16, _lib.FN_F16])
@pytest.mark.parametrize("case", TAPS_CASES)
def test_conv_wgrad_tap_sharing_kernel(lib, case, dt):
    """conv_wgrad_taps_kernel (k x k layers on maps of >= 32 pixels; grouped path only): dW against an fp32 CPU convolution
    gradient on the same rounded operands -- unsplit (direct stores), split over pixels (slabs + ordered reduce), and
    bit-identical from run to run."""
    N, H, W, Cin, Cout, kh, kw, s, ph, pw, ld_x, ld_y, variant = case
    d = conv_desc(N, H, W, Cin, Cout, kh, kw, s, ph, pw, dt, ld_x=ld_x, ld_y=ld_y)
    xb = _mk((N, H, W, ld_x), dt, seed=61)
    dyb = _mk((N, d.OH, d.OW, ld_y), dt, seed=62)
    x0, y0 = (ld_x - Cin) // 2 // 8 * 8, (ld_y - Cout) // 2 // 8 * 8      # the slice starts inside the buffer
    x, dy = xb[..., x0:x0 + Cin], dyb[..., y0:y0 + Cout]
    dw = torch.full((Cout, kh, kw, Cin), float("nan"), dtype=torch.float32, device="cuda")
    d.x, d.y, d.dw = ptr(xb, x0), ptr(dyb, y0), ptr(dw)
    assert lib.fn_conv2d_variant(C.byref(d), 2) == variant
    wr = torch.zeros(Cout, Cin, kh, kw, requires_grad=True)
    yr = torch.nn.functional.conv2d(x.float().cpu().permute(0, 3, 1, 2), wr, None, stride=s, padding=(ph, pw))
    yr.backward(dy.float().cpu().permute(0, 3, 1, 2))
    ref = wr.grad.permute(0, 2, 3, 1)
    nbytes = lib.fn_conv2d_wgrad_arg_bytes()
    for splits in (1, 3, 0):                       # 0: the library's own choice
        d.splits = splits
        arr = (_lib.ConvDesc * 1)(d)
        host_args, host_prefix, ws_elems = (C.c_uint8 * nbytes)(), (C.c_int32 * 2)(), C.c_int64(0)
        total = lib.fn_conv2d_wgrad_group_build(arr, 1, variant, host_args, host_prefix, None, C.byref(ws_elems))
        nslabs = ws_elems.value // dw.numel()        # a layer cannot be split into more pieces than it has pixel tiles
        assert total > 0 and ws_elems.value % dw.numel() == 0 and (splits != 1 or nslabs == 0) and (splits != 3 or 2 <= nslabs <= 3)
        ws = torch.full((max(1, ws_elems.value),), float("nan"), device="cuda")
        total = lib.fn_conv2d_wgrad_group_build(arr, 1, variant, host_args, host_prefix, ptr(ws), C.byref(ws_elems))
        dev_args = torch.frombuffer(bytearray(host_args), dtype=torch.uint8).cuda()
        dev_prefix = torch.tensor(list(host_prefix), dtype=torch.int32, device="cuda")
        runs = []
        for _ in range(2):
            dw.fill_(float("nan"))
            _lib.check(lib.fn_conv2d_wgrad_grouped(ptr(dev_args), ptr(dev_prefix), 1, total, variant, dt, stream()))
            _lib.check(lib.fn_conv2d_wgrad_reduce(ptr(dev_args), 1, stream()))
            torch.cuda.synchronize()
            runs.append(dw.clone())
        assert rel_err(runs[0], ref) < 2e-5, splits
        assert torch.equal(runs[0], runs[1])
    # the single-layer launch of the same descriptor (general kernel, atomics) agrees
    d.splits = 0
    dw.zero_()
    _lib.check(lib.fn_conv2d_wgrad(C.byref(d), stream()))
    torch.cuda.synchronize()
    assert rel_err(dw, runs[0]) < 2e-5
    # layers the kernel does not take: 3x3 maps, fewer than 32 input channels, 1x1
    for (hh, cin, k) in ((3, 192, 3), (19, 8, 3), (17, 64, 1)):
        e = conv_desc(2, hh, hh, cin, 64, k, k, 1, k // 2, k // 2, dt)
        e.x = e.y = e.dw = ptr(dw)
        assert lib.fn_conv2d_variant(C.byref(e), 2) < 5000000


def test_conv_rejects_bad_geometry(lib):
    d = conv_desc(1, 8, 8, 12, 16, 3, 3, 1, 0, 0, _lib.FN_BF16)   # Cin not a multiple of 8
    x = torch.zeros(1, 8, 8, 12, dtype=torch.bfloat16, device="cuda")
    d.x = d.w = d.y = ptr(x)
    with pytest.raises(ValueError):
        _lib.check(lib.fn_conv2d_fwd(C.byref(d), stream()))


def test_conv_wgrad_grouped_matches_single_launches(lib):
    """fn_conv2d_wgrad_grouped: several layers of one tile variant in ONE launch == the per-layer launches."""
    dt = _lib.FN_BF16
    cases = [(3, 8, 8, 128, 128, 1, 7, 1, 0, 3), (3, 8, 8, 896, 128, 1, 1, 1, 0, 0), (2, 17, 17, 192, 192, 3, 3, 1, 1, 1),
             (5, 3, 3, 192, 192, 3, 1, 1, 1, 0), (2, 17, 17, 192, 256, 3, 3, 2, 0, 0),
             # second and third members for every tap-sharing variant (records of a group are fn_conv2d_wgrad_arg_bytes() apart)
             (2, 19, 19, 64, 64, 3, 3, 1, 0, 0), (1, 33, 21, 96, 128, 3, 3, 1, 1, 1), (2, 17, 17, 64, 96, 3, 3, 2, 0, 0),
             (3, 8, 8, 128, 128, 7, 1, 1, 3, 0), (2, 9, 12, 64, 80, 1, 7, 1, 0, 3), (4, 17, 17, 256, 32, 1, 1, 1, 0, 0)]
    descs, keep, singles = [], [], []
    for i, (N, H, W, Cin, Cout, kh, kw, s, ph, pw) in enumerate(cases):
        d = conv_desc(N, H, W, Cin, Cout, kh, kw, s, ph, pw, dt)
        x = _mk((N, H, W, Cin), dt, seed=31 + i)
        dy = _mk((N, d.OH, d.OW, Cout), dt, seed=41 + i)
        dw = torch.zeros(Cout, kh, kw, Cin, dtype=torch.float32, device="cuda")
        ref = torch.zeros_like(dw)
        d.x, d.y, d.dw = ptr(x), ptr(dy), ptr(ref)
        _lib.check(lib.fn_conv2d_wgrad(C.byref(d), stream()))
        d.dw = ptr(dw)
        descs.append(d); keep.append((x, dy, dw, ref))
    groups = {}
    for d, k in zip(descs, keep):
        groups.setdefault(lib.fn_conv2d_variant(C.byref(d), 2), []).append((d, k))
    assert len(groups) >= 1 and [len(m) for v, m in groups.items() if v >= 5000000] == [8]
    nbytes = lib.fn_conv2d_wgrad_arg_bytes()
    for variant, members in groups.items():
        n = len(members)
        arr = (_lib.ConvDesc * n)(*[m[0] for m in members])
        host_args = (C.c_uint8 * (nbytes * n))()
        host_prefix = (C.c_int32 * (n + 1))()
        ws_elems = C.c_int64(0)
        assert lib.fn_conv2d_wgrad_group_build(arr, n, variant, host_args, host_prefix, None, C.byref(ws_elems)) > 0     # sizing call
        ws = torch.full((max(1, ws_elems.value),), float("nan"), device="cuda")           # every slab element must be written
        total = lib.fn_conv2d_wgrad_group_build(arr, n, variant, host_args, host_prefix, ptr(ws), C.byref(ws_elems))
        assert total > 0 and list(host_prefix)[0] == 0 and list(host_prefix)[-1] == total
        dev_args = torch.frombuffer(bytearray(host_args), dtype=torch.uint8).cuda()
        dev_prefix = torch.tensor(list(host_prefix), dtype=torch.int32, device="cuda")
        for _, k in members:
            k[2].fill_(float("nan"))                                                       # dw needs no zeroing on the grouped path
        _lib.check(lib.fn_conv2d_wgrad_grouped(ptr(dev_args), ptr(dev_prefix), n, total, variant, dt, stream()))
        _lib.check(lib.fn_conv2d_wgrad_reduce(ptr(dev_args), n, stream()))
        torch.cuda.synchronize()
        first = [k[2].clone() for _, k in members]
        assert not any(bool(torch.isnan(f).any()) for f in first)
        for _ in range(3):                      # atomic-free, fixed summation order: the same bits every run
            _lib.check(lib.fn_conv2d_wgrad_grouped(ptr(dev_args), ptr(dev_prefix), n, total, variant, dt, stream()))
            _lib.check(lib.fn_conv2d_wgrad_reduce(ptr(dev_args), n, stream()))
            torch.cuda.synchronize()
            assert all(torch.equal(f, k[2]) for f, (_, k) in zip(first, members))
    torch.cuda.synchronize()
    for (x, dy, dw, ref) in keep:
        assert rel_err(dw, ref) < 2e-5
    # a descriptor of another variant is rejected
    bad = (_lib.ConvDesc * 1)(descs[0])
    other = [v for v in (32064, 64064, 128128) if v != lib.fn_conv2d_variant(C.byref(descs[0]), 2)][0]
    with pytest.raises(ValueError):
        _lib.check(lib.fn_conv2d_wgrad_group_build(bad, 1, other, (C.c_uint8 * nbytes)(), (C.c_int32 * 2)(), None, C.byref(C.c_int64(0))))


@pytest.mark.parametrize("dt", [_lib.FN_BF16, _lib.FN_F16])
@pytest.mark.parametrize("H", [17, 37])           # 37: the halo-tile kernel runs the same shared epilogue
def test_dgrad_fused_bn_backward_reduction(lib, dt, H):
    """dgrad epilogue computing sum(dyh) / sum(dyh*xhat) of the producing layer's BatchNorm (+ReLU) == the standalone
    reduce kernel; the apply kernel then consumes the replicated accumulators."""
    N, W, Cin, Cout = 3, H, 64, 96
    d = conv_desc(N, H, W, Cin, Cout, 3, 3, 1, 1, 1, dt)
    if H == 37:
        d.tile_dgrad = HALO          # 96 source channels: production would take the implicit-GEMM kernel
    M = N * H * W
    dy = _mk((N, H, W, Cout), dt, seed=51)
    w = _mk((Cout, 3, 3, Cin), dt, 0.1, seed=52)
    wt = torch.zeros_like(w).view(-1)
    table = torch.tensor([[0, Cout, 9 * Cin, 9, Cin, -1, -1, 0]], dtype=torch.int32, device="cuda")
    _lib.check(lib.fn_pack_transpose(ptr(w), ptr(wt), ptr(table), 1, w.numel(), dt, stream()))
    yraw = _mk((N, H, W, Cin), dt, seed=53, scale=2.0)
    beta = (torch.randn(Cin, generator=torch.Generator().manual_seed(54)) * 0.3).cuda()
    yf = yraw.float().view(M, Cin)
    mean, var = yf.mean(0), yf.var(0, unbiased=False)
    sc = torch.rsqrt(var + 1e-3).contiguous()
    sh = (beta - mean * sc).contiguous()
    reps = 4
    acc = torch.zeros(reps, 2 * Cin, dtype=torch.int64, device="cuda")      # fixed point, FN_ACC_GRAD_BITS
    dx = torch.zeros(N, H, W, Cin, dtype=lp_dtype(dt), device="cuda")
    d.y, d.w, d.dx = ptr(dy), ptr(wt), ptr(dx)
    d.bn_y, d.ld_bn_y, d.bn_scale, d.bn_shift, d.bn_beta = ptr(yraw), Cin, ptr(sc), ptr(sh), ptr(beta)
    d.bn_acc, d.bn_sq_off, d.bn_replicas, d.bn_rep_stride, d.bn_relu = ptr(acc), Cin, reps, 2 * Cin, 1
    assert (lib.fn_conv2d_variant(C.byref(d), 1) >= 9000000) == (H == 37)
    _lib.check(lib.fn_conv2d_dgrad(C.byref(d), stream()))
    # reference: standalone reduce on the (rounded) dx
    acc_ref = torch.zeros(2 * Cin, dtype=torch.int64, device="cuda")
    dbeta_ref = torch.zeros(Cin, device="cuda")
    dx_ref = dx.clone()
    _lib.check(lib.fn_bn_relu_train_bwd(ptr(dx_ref), Cin, ptr(yraw), Cin, M, Cin, ptr(beta), ptr(sc), ptr(sh), ptr(dbeta_ref), ptr(acc_ref), Cin, 1, 0,
                                        0, 1, dt, stream()))
    dbeta = torch.zeros(Cin, device="cuda")
    _lib.check(lib.fn_bn_relu_train_bwd(ptr(dx), Cin, ptr(yraw), Cin, M, Cin, ptr(beta), ptr(sc), ptr(sh), ptr(dbeta), ptr(acc), Cin, reps, 2 * Cin,
                                        1, 1, dt, stream()))
    torch.cuda.synchronize()
    tol = 2e-2 if dt == _lib.FN_BF16 else 3e-3       # fused sums use the un-rounded fp32 gradient
    assert rel_err(from_acc(acc.sum(0), ACC_GRAD_BITS), from_acc(acc_ref, ACC_GRAD_BITS)) < tol
    assert rel_err(dbeta, dbeta_ref) < tol
    assert rel_err(dx, dx_ref) < tol


NORM_CASES = [c for c in CASES if c[3] <= 512 and c[3] >= 32]


@pytest.mark.parametrize("dt", [_lib.FN_BF16, _lib.FN_F16])
@pytest.mark.parametrize("case", NORM_CASES)
def test_conv_normalise_on_load_equals_materialised_bn(lib, case, dt):
    """fwd and wgrad with nrm_* on the RAW tensor must give exactly what they give on the tensor fn_bn_relu_train_fwd writes
    (same statistics, same rounding of the activated operand), including zero padding and channel slices."""
    N, H, W, Cin, Cout, kh, kw, s, ph, pw = case
    M = N * H * W
    ld = Cin + 16                                      # x is the slice [8, 8+Cin) of a wider buffer
    raw = _mk((N, H, W, ld), dt, 1.5, seed=11)
    rawf = raw[..., 8:8 + Cin].float().reshape(M, Cin)
    reps, CBs = 4, Cin + 24                            # statistics live at offset 16 of a wider channel space
    stats = torch.zeros(reps, 2 * CBs, dtype=torch.int64, device="cuda")
    part = torch.arange(M, device="cuda") % reps
    for r in range(reps):                              # replicas hold partial sums (fixed point), as the producing conv leaves them
        stats[r, 16:16 + Cin] = to_acc(rawf[part == r].sum(0), ACC_STAT_BITS)
        stats[r, CBs + 16:CBs + 16 + Cin] = to_acc((rawf[part == r] ** 2).sum(0), ACC_STAT_BITS)
    beta = (torch.randn(CBs, generator=torch.Generator().manual_seed(5)) * 0.3).cuda()
    z = torch.zeros_like(raw)
    sc, sh = torch.zeros(CBs, device="cuda"), torch.zeros(CBs, device="cuda")
    _lib.check(lib.fn_bn_relu_train_fwd(ptr(raw, 8), ld, ptr(z, 8), ld, M, Cin, ptr(stats, 16), CBs, reps, 2 * CBs, ptr(beta, 16),
                                        ptr(sc, 16), ptr(sh, 16), None, None, 0.99, 1e-3, 1, dt, stream()))
    w = _mk((Cout, kh, kw, Cin), dt, 0.1, seed=12)
    outs = []
    for norm in (False, True):
        d = conv_desc(N, H, W, Cin, Cout, kh, kw, s, ph, pw, dt, ld_x=ld)
        y = torch.zeros(N, d.OH, d.OW, Cout, dtype=lp_dtype(dt), device="cuda")
        d.w, d.y = ptr(w), ptr(y)
        if norm:
            d.x, d.nrm_stats, d.nrm_beta = ptr(raw, 8), ptr(stats, 16), ptr(beta, 16)
            d.nrm_sq_off, d.nrm_replicas, d.nrm_rep_stride, d.nrm_count, d.nrm_eps = CBs, reps, 2 * CBs, M, 1e-3
        else:
            d.x = ptr(z, 8)
        _lib.check(lib.fn_conv2d_fwd(C.byref(d), stream()))
        dy = _mk((N, d.OH, d.OW, Cout), dt, seed=13)
        dw = torch.zeros(Cout, kh * kw * Cin, dtype=torch.float32, device="cuda")
        d.y, d.dw, d.splits = ptr(dy), ptr(dw), 1      # one split: the accumulation order is fixed, results comparable bit for bit
        _lib.check(lib.fn_conv2d_wgrad(C.byref(d), stream()))
        torch.cuda.synchronize()
        outs.append((y, dw))
    # operands are bit-identical; the plain launch may use in-launch split-K (another summation order), the nrm one never does
    assert rel_err(outs[1][0], outs[0][0]) < (2e-3 if dt == _lib.FN_BF16 else 3e-4)
    assert torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][0].float().abs().max()) > 0 and float(outs[0][1].abs().max()) > 0
    # fn_bn_finalize publishes the same scale / shift and moving statistics as the materialising kernel
    sc2, sh2 = torch.zeros(CBs, device="cuda"), torch.zeros(CBs, device="cuda")
    mm, mv = torch.zeros(CBs, device="cuda"), torch.ones(CBs, device="cuda")
    mm_ref, mv_ref = torch.zeros(CBs, device="cuda"), torch.ones(CBs, device="cuda")
    _lib.check(lib.fn_bn_relu_train_fwd(ptr(raw, 8), ld, ptr(z, 8), ld, M, Cin, ptr(stats, 16), CBs, reps, 2 * CBs, ptr(beta, 16),
                                        ptr(sc, 16), ptr(sh, 16), ptr(mm_ref, 16), ptr(mv_ref, 16), 0.99, 1e-3, 1, dt, stream()))
    fr = torch.zeros(CBs, dtype=torch.int32, device="cuda")
    fr[16:16 + Cin] = reps
    fc = torch.full((CBs,), M, dtype=torch.int32, device="cuda")
    _lib.check(lib.fn_bn_finalize(ptr(stats), CBs, 2 * CBs, ptr(fr), ptr(fc), ptr(beta), ptr(sc2), ptr(sh2), ptr(mm), ptr(mv), 0.99, 1e-3, CBs,
                                  stream()))
    torch.cuda.synchronize()
    assert torch.equal(sc, sc2) and torch.equal(sh, sh2) and torch.equal(mm, mm_ref) and torch.equal(mv, mv_ref)
    assert float(sc2[:16].abs().max()) == 0 and float(sc2[16 + Cin:].abs().max()) == 0      # channels with reps == 0 untouched


def test_conv_normalise_on_load_rejects_wide_inputs(lib):
    d = conv_desc(1, 3, 3, 1792, 128, 1, 1, 1, 0, 0, _lib.FN_BF16)
    x = torch.zeros(1, 3, 3, 1792, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(128, 1, 1, 1792, dtype=torch.bfloat16, device="cuda")
    y = torch.zeros(1, 3, 3, 128, dtype=torch.bfloat16, device="cuda")
    st = torch.zeros(2 * 1792, dtype=torch.int64, device="cuda")
    d.x, d.w, d.y, d.nrm_stats, d.nrm_beta, d.nrm_count, d.nrm_eps, d.nrm_sq_off = ptr(x), ptr(w), ptr(y), ptr(st), ptr(st), 9, 1e-3, 1792
    with pytest.raises(ValueError):
        _lib.check(lib.fn_conv2d_fwd(C.byref(d), stream()))


@pytest.mark.parametrize("dt", [_lib.FN_BF16, _lib.FN_F16])
@pytest.mark.parametrize("nsrc", [2, 3])
@pytest.mark.parametrize("tile", [0, 128064, 64064, 32032, 32128])
def test_dgrad_sibling_sources_equal_the_sum_of_single_dgrads(lib, dt, nsrc, tile):
    """dX of 2-3 sibling 1x1 layers as ONE multi-source launch (dy2/w2, dy3/w3, own row strides and widths, K tails that are
    not multiples of the 64-wide k tile) == the fp32 sum of the per-layer products, with and without accumulation."""
    N, H, W, Cin = 3, 9, 9, 256
    couts, lds = [32, 40, 96][:nsrc], [96, 40, 160][:nsrc]          # dY slices of wider buffers (ld > Cout) like `mixed`
    M = N * H * W
    dys = [_mk((N, H, W, ld), dt, seed=20 + i) for i, ld in enumerate(lds)]
    wts = [_mk((Cin, 1, 1, c), dt, 0.1, seed=30 + i) for i, c in enumerate(couts)]      # transposed packs [Cin][tap][Cout]
    ref = sum(dy[..., :c].float().cpu().reshape(M, c) @ wt.float().cpu().reshape(Cin, c).t() for dy, wt, c in zip(dys, wts, couts))
    d = conv_desc(N, H, W, Cin, couts[0], 1, 1, 1, 0, 0, dt, ld_y=lds[0])
    d.tile_dgrad = tile
    d.y, d.w = ptr(dys[0]), ptr(wts[0])
    d.dy2, d.w2, d.Cout2, d.ld_y2 = ptr(dys[1]), ptr(wts[1]), couts[1], lds[1]
    if nsrc == 3:
        d.dy3, d.w3, d.Cout3, d.ld_y3 = ptr(dys[2]), ptr(wts[2]), couts[2], lds[2]
    dx = torch.full((N, H, W, Cin), 3.0, dtype=lp_dtype(dt), device="cuda")
    d.dx = ptr(dx)
    _lib.check(lib.fn_conv2d_dgrad(C.byref(d), stream()))
    torch.cuda.synchronize()
    tol = 6e-3 if dt == _lib.FN_BF16 else 8e-4
    assert rel_err(dx.reshape(M, Cin), ref) < tol
    base = dx.float().cpu().reshape(M, Cin).clone()
    d.accumulate = 1
    _lib.check(lib.fn_conv2d_dgrad(C.byref(d), stream()))
    torch.cuda.synchronize()
    assert rel_err(dx.reshape(M, Cin), base + ref) < 2 * tol
    d.Cout2 = 33                                                    # not a multiple of 8
    with pytest.raises(ValueError):
        _lib.check(lib.fn_conv2d_dgrad(C.byref(d), stream()))
