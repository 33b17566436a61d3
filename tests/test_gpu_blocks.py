"""Block-level forward + backward parity ("block: Block35/17/8, ReductionA/B fwd+bwd vs CPU restatement").

Each residual / reduction block of facenet/models/inception_resnet_v1.py:83-377 is lowered ALONE by the engine
(engine.BlockNetwork: the same lowering code, the same kernels, the same fusions as in the full network) at batch >= 32 and
compared with fp32 autograd on identically rounded operands: tests/quant_oracle.py rounds weights, raw convolution outputs,
BN+ReLU outputs and the block output to the storage type at the points the HIP path stores them (straight-through
gradients), everything else is PyTorch-CPU fp32.  Tolerances (relative L2): output 4e-3 / 1e-3, dX and every dW / dbeta / dbias
1e-2 (bf16) and 2e-3 (f16).  PARITY UNPINNED by the reference (no vectors)."""
import numpy as np
import pytest
import torch

from facenet_amd.engine import BLOCK_TOWERS, BlockNetwork
from oracle import facenet_oracle as fo
from tests.quant_oracle import QuantOracle

pytestmark = pytest.mark.gpu

CASES = [  # kind, H, W, C, N, scale, relu, repeat
    ("block35", 17, 17, 256, 32, 0.17, True, 1),
    ("block17", 8, 8, 896, 32, 0.10, True, 1),
    ("block17", 8, 8, 896, 32, 0.10, True, 3),   # a chain: the residual backward of block i runs in the epilogue of block i+1's data gradient
    ("block8", 3, 3, 1792, 48, 0.2, True, 1),
    ("block8", 3, 3, 1792, 48, 1.0, False, 1),   # the last Block8: scale 1, no activation (:453)
    ("reduction_a", 17, 17, 256, 32, 0.0, True, 1),
    ("reduction_b", 8, 8, 896, 32, 0.0, True, 1),
]


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def _oracle_block(params, kind, x_nchw, scale, relu, dt, repeat=1, masks=None):
    o = QuantOracle(params, dt, masks=masks)
    if kind in BLOCK_TOWERS:
        blk = {"block35": fo.BLOCK35, "block17": fo.BLOCK17, "block8": fo.BLOCK8}[kind]
        for i in range(repeat):
            x_nchw = o._block(x_nchw, f"{kind}/{i}", blk, scale, "relu" if relu else None, True)
        return x_nchw, o
    spec = fo.reduction_a_spec(fo.DEFAULT_CONFIG["reduction_a"]["filters"]) if kind == "reduction_a" else \
        fo.reduction_b_spec(fo.DEFAULT_CONFIG["reduction_b"]["filters"])
    return o._reduction(x_nchw, kind, spec, True), o


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("kind,H,W,C,N,scale,relu,repeat", CASES)
def test_block_forward_backward_against_rounded_autograd(kind, H, W, C, N, scale, relu, repeat, dt):
    net = BlockNetwork(kind, H, W, C, scale=scale, relu=relu, repeat=repeat, device="cuda:0", train_dtype=dt, seed=11)
    # non-trivial beta / bias so that their gradients and the ReLU masks are exercised
    g = torch.Generator().manual_seed(5)
    params = net.export_keras_params()
    for k in params:
        if k.endswith("/beta") or k.endswith("/bias"):
            params[k] = 0.1 * torch.randn(params[k].shape, generator=g)
    net.load_keras_params(params)
    trainable = [k for k in params if not k.endswith(("/moving_mean", "/moving_variance"))]
    plan = net.plan(N, training=True)
    net.alloc_grads()
    # the trunk is the output of a ReLU in the real network: non-negative, O(1)
    x = torch.relu(torch.randn(N, H, W, C, generator=g) + 0.3).to(dt)
    trunk, out = plan.bufs["trunk"], plan.embedding.buf
    trunk.act.copy_(x)
    st = net.stream()
    plan.ws.zero_(); plan.ws_b.zero_()
    plan.run_ops(plan.fwd, st)
    dout = (0.05 * torch.randn(N, out.H, out.W, out.C, generator=g)).to(dt)
    out.grad.copy_(dout)
    plan.build_backward(None)
    plan.run_ops(plan.bwd, st)
    torch.cuda.synchronize()

    # reference: fp32 autograd, operands rounded where the HIP path stores them.  Two passes: (1) the oracle's own ReLU active
    # sets -- the forward must agree, and the gradient distance is then bounded by the sign flips of near-zero pre-activations
    # (~sqrt of the forward distance, see quant_oracle._ReluWithMask); (2) the device's active sets -- what remains is the
    # backward arithmetic, held to 2e-3 (f16) / 1e-2 (bf16).
    masks = {}
    for r in plan.recs:
        if r.kind == "bn":
            for r2 in plan.recs:                  # the BN range covers the output slices of one or more convolutions
                if r2.kind == "conv" and r2.y.buf is r.y.buf and r2.extra.get("kind") == "bn":
                    a = r2.y.buf.act[..., r2.y.c0:r2.y.c0 + r2.y.C]
                    masks[r2.layer.name] = (a > 0).float().cpu().permute(0, 3, 1, 2).contiguous()
        elif r.kind == "conv" and r.extra.get("kind") == "resid" and relu:
            masks[r.layer.name[:-3]] = (r.y.buf.act > 0).float().cpu().permute(0, 3, 1, 2).contiguous()     # "<block>/up" -> "<block>"
    mine = net.export_keras_grads(net.G)
    tol_y, tol_g = (1e-3, 2e-3) if dt == torch.float16 else (4e-3, 1e-2)
    for shared in (False, True):
        for k in trainable:
            params[k].requires_grad_(True)
            params[k].grad = None
        xr = x.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        y_ref, orc = _oracle_block(params, kind, xr, scale, relu, dt, repeat, masks if shared else None)
        y_ref.backward(dout.float().permute(0, 3, 1, 2))
        y_ref = y_ref.detach().permute(0, 2, 3, 1)
        dx_ref = xr.grad.permute(0, 2, 3, 1)
        e_y = _rel(out.act.float().cpu(), y_ref)
        e_dx = _rel(trunk.grad.float().cpu(), dx_ref)
        errs = {k: _rel(mine[k], params[k].grad) for k in trainable if params[k].grad is not None and params[k].grad.norm() > 1e-6}
        worst = max(errs, key=errs.get)
        print(f"{kind} x{repeat} N={N} {dt} {'device masks' if shared else 'own masks   '}: out {e_y:.2e}  dX {e_dx:.2e}  "
              f"worst dW {worst} {errs[worst]:.2e}  median {np.median(list(errs.values())):.2e}")
        assert e_y < tol_y
        assert len(errs) >= len(net.layers)          # every kernel has a gradient
        bound = tol_g if shared else 10 * np.sqrt(e_y) + tol_g
        assert e_dx < bound
        for k, e in errs.items():
            assert e < bound, (k, e, shared)
    # moving statistics: momentum 0.99, biased batch variance (hazard 3), from the un-rounded accumulators
    new = net.export_keras_params()
    assert orc.new_stats, "the oracle recorded no moving statistics"
    for k, v in orc.new_stats.items():
        assert torch.allclose(new[k], v, rtol=2e-2, atol=2e-3), k


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_batchnorm_statistics_with_mean_far_from_zero(dt, lib):
    """Keras' non-fused BatchNormalization (inception_resnet_v1.py:56-63) takes the variance as mean((x - mean)^2); the
    convolution epilogue accumulates sum and sum of squares in fp32 (var = E[x^2] - mean^2), whose cancellation error grows
    as (mean/std)^2 * 1e-7.  Channels with |mean| = 50 std must give the two-pass statistics of an fp64 reference: rstd within
    1e-3 relative, every normalised value within 5e-3 of (x - mean) * rstd.  At |mean| = 200 std -- far outside anything a
    zero-mean-initialised, BatchNorm-centred network produces -- the error is bounded at 1 % (documented limit)."""
    import ctypes as C
    from facenet_amd import _lib
    from tests.util import conv_desc, ptr, stream
    code = _lib.dtype_code(dt)
    N, H, W, Cc = 32, 35, 35, 64
    M = N * H * W
    g = torch.Generator().manual_seed(0)
    ratio = torch.tensor([50.0] * 32 + [200.0] * 16 + [0.0] * 16)
    std = torch.tensor([1.0] * 16 + [0.05] * 16 + [0.25] * 32)
    x = (torch.randn(M, Cc, generator=g) * std + ratio * std).to(dt)          # what the kernel reads (already rounded)
    w = torch.eye(Cc).to(dt)                                                    # 1x1 identity: y == x exactly
    dev = "cuda:0"
    xd, wd = x.to(dev), w.to(dev)
    y = torch.zeros(M, Cc, dtype=dt, device=dev)
    z = torch.zeros(M, Cc, dtype=dt, device=dev)
    reps = 16
    ws = torch.zeros(reps * 2 * Cc, dtype=torch.int64, device=dev)          # fixed-point accumulators (fn_acc_t)
    d = conv_desc(N, H, W, Cc, Cc, 1, 1, 1, 0, 0, code)
    d.x, d.w, d.y = ptr(xd), ptr(wd), ptr(y)
    d.stats, d.stats_sq_off, d.stats_replicas, d.stats_rep_stride = ptr(ws), Cc, reps, 2 * Cc
    _lib.check(lib.fn_conv2d_fwd(C.byref(d), stream()), "conv_fwd")
    beta = torch.zeros(Cc, device=dev)
    sc, sh = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
    mm, mv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
    _lib.check(lib.fn_bn_relu_train_fwd(ptr(y), Cc, ptr(z), Cc, M, Cc, ptr(ws), Cc, reps, 2 * Cc, ptr(beta), ptr(sc), ptr(sh), ptr(mm), ptr(mv),
                                        0.99, 1e-3, 0, code, stream()), "bn_fwd")
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), x)
    x64 = x.double()
    mean = x64.mean(0)
    var = ((x64 - mean) ** 2).mean(0)
    rstd = 1.0 / torch.sqrt(var + 1e-3)
    r_rstd = (sc.cpu().double() - rstd).abs() / rstd
    # what the consumers see: z = x*scale + shift against the two-pass (x - mean)*rstd, before storage rounding
    z_dev = x64 * sc.cpu().double() + sh.cpu().double()
    e_z = (z_dev - (x64 - mean) * rstd).abs().amax(0)
    e_mv = ((mv.cpu().double() - (0.99 + 0.01 * var)).abs() / (0.99 + 0.01 * var)).max().item()
    r50, r200, r0 = slice(0, 32), slice(32, 48), slice(48, 64)
    print(f"{dt}: rstd rel err |mean|=50std {r_rstd[r50].max():.2e}, 200std {r_rstd[r200].max():.2e}, 0 {r_rstd[r0].max():.2e}; "
          f"z abs err {e_z[r50].max():.2e} / {e_z[r200].max():.2e} / {e_z[r0].max():.2e}; moving var rel err {e_mv:.2e}")
    assert r_rstd[r50].max() < 1e-3 and r_rstd[r0].max() < 1e-5
    assert e_z[r50].max() < 5e-3 and e_z[r0].max() < 1e-4
    assert r_rstd[r200].max() < 1e-2 and e_z[r200].max() < 5e-2
    assert e_mv < 1e-3


def test_block17_warm_ahead_entry_point(lib):
    """fn_block17_infer_warm: the warm-ahead workgroups (block index >= N: they only read [warm, warm + warm_bytes) into L2) change no bit
    of the output, whatever the range -- including one that is no multiple of the 8 x 16-byte eighths -- and a range without a size, a
    misaligned range or a negative size is FN_EINVAL with nothing launched."""
    from facenet_amd import _lib
    from tests.util import ptr, stream
    torch.manual_seed(3)
    N, dt = 5, torch.float16
    x = (torch.randn(N, 8, 8, 896, device="cuda") * 0.5).to(dt)
    ws = [(torch.randn(*s, device="cuda") * 0.03).to(dt) for s in ((128, 896), (128, 896), (128, 7, 128), (128, 7, 128), (896, 256))]
    bs = [torch.randn(n, device="cuda") * 0.1 for n in (128, 128, 128, 128, 896)]
    junk = torch.randn(70001, device="cuda").to(dt)                     # 140 002 bytes: ragged against every chunking

    def run(warm, nbytes):
        y = torch.zeros_like(x)
        _lib.check(lib.fn_block17_infer_warm(ptr(x), ptr(y), N, *[ptr(w) for w in ws], *[ptr(b) for b in bs], 0.1, 1, warm, nbytes,
                                              _lib.FN_F16, stream()))
        torch.cuda.synchronize()
        return y

    ref = torch.zeros_like(x)
    _lib.check(lib.fn_block17_infer(ptr(x), ptr(ref), N, *[ptr(w) for w in ws], *[ptr(b) for b in bs], 0.1, 1, _lib.FN_F16, stream()))
    torch.cuda.synchronize()
    assert float(ref.float().abs().max()) > 0
    assert torch.equal(run(None, 0), ref)
    assert torch.equal(run(ptr(junk), 140000 // 16 * 16), ref)
    assert torch.equal(run(ptr(ws[4]), ws[4].numel() * 2), ref)
    assert torch.equal(run(ptr(junk), 16), ref)
    for warm, nbytes in ((ptr(junk), 0), (None, 64), (ptr(junk) + 2, 64), (ptr(junk), -16)):
        y = torch.zeros_like(x)
        with pytest.raises(ValueError):
            _lib.check(lib.fn_block17_infer_warm(ptr(x), ptr(y), N, *[ptr(w) for w in ws], *[ptr(b) for b in bs], 0.1, 1, warm, nbytes,
                                                  _lib.FN_F16, stream()))
        torch.cuda.synchronize()
        assert float(y.float().abs().max()) == 0.0                      # nothing was launched
    assert b"warm" in lib.fn_last_error()
