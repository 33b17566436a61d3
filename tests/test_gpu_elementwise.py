"""HBM-bound kernels (normalisation, BN fwd/bwd, pools, residual bwd, head, losses, Adam) against PyTorch-CPU fp32."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from facenet_amd import _lib
from oracle import facenet_oracle as fo
from tests.util import ACC_GRAD_BITS, ACC_STAT_BITS, from_acc, to_acc, lp_dtype, ptr, rel_err, stream

pytestmark = pytest.mark.gpu
BF, HF = _lib.FN_BF16, _lib.FN_F16


def _rand(shape, dt=None, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    t = torch.randn(shape, generator=g) * scale
    return t.to(lp_dtype(dt)).cuda() if dt is not None else t.cuda()


@pytest.mark.parametrize("mode", [0, 1])
def test_image_normalize(lib, mode):
    x = np.random.default_rng(0).integers(0, 256, (3, 160, 160, 3), dtype=np.uint8)
    x[1] = 77                       # constant image: range clamps to eps (facenet.py:76)
    xt = torch.from_numpy(x).cuda()
    out = torch.zeros(3, 160, 160, 8, dtype=torch.float16, device="cuda")
    work = torch.zeros(24, dtype=torch.float32, device="cuda")      # 8 words per image
    _lib.check(lib.fn_image_normalize(ptr(xt), ptr(out), ptr(work), 3, 160 * 160, mode, HF, stream()))
    ref = fo.image_processing(x, normalization=mode)
    assert torch.allclose(out[..., :3].float().cpu(), ref, atol=2e-3, rtol=2e-3)
    assert float(out[..., 3:].abs().max()) == 0
    # float input path (facenet.py:69 casts anyway)
    xf = torch.from_numpy(x.astype(np.float32)).cuda()
    out2 = torch.zeros_like(out)
    _lib.check(lib.fn_image_normalize_f32(ptr(xf), ptr(out2), ptr(work), 3, 160 * 160, mode, HF, stream()))
    assert torch.equal(out, out2)
    with pytest.raises(ValueError):
        _lib.check(lib.fn_image_normalize(ptr(xt), ptr(out), ptr(work), 3, 160 * 160, 2, HF, stream()))


@pytest.mark.parametrize("dt", [BF, HF])
@pytest.mark.parametrize("shape", [(700, 32, 96, 32), (333, 80, 80, 0), (64, 896, 1024, 64)])
def test_bn_relu_train_fwd_bwd(lib, dt, shape):
    M, Cc, ld, c0 = shape
    ybuf = _rand((M, ld), dt, seed=1, scale=2.0)
    y = ybuf[:, c0:c0 + Cc]
    beta = (torch.randn(Cc, generator=torch.Generator().manual_seed(2)) * 0.3).cuda()
    yf = y.float()
    stats = to_acc(torch.cat([yf.sum(0), (yf * yf).sum(0)]), ACC_STAT_BITS).contiguous()     # fixed-point accumulators (fn_acc_t)
    z = torch.zeros(M, ld, dtype=lp_dtype(dt), device="cuda")
    sc, sh = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
    mm, mv = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
    _lib.check(lib.fn_bn_relu_train_fwd(ptr(ybuf, c0), ld, ptr(z, c0), ld, M, Cc, ptr(stats), Cc, 1, 0, ptr(beta), ptr(sc), ptr(sh), ptr(mm), ptr(mv),
                                        0.99, 1e-3, 1, dt, stream()))
    yr = yf.cpu().clone().requires_grad_(True)
    mean, var = yr.mean(0), yr.var(0, unbiased=False)
    zr = F.relu((yr - mean) * torch.rsqrt(var + 1e-3) + beta.cpu())
    assert rel_err(z[:, c0:c0 + Cc], zr.detach()) < (5e-3 if dt == BF else 6e-4)
    assert torch.allclose(mm.cpu(), 0.01 * mean.detach(), atol=1e-5)
    assert torch.allclose(mv.cpu(), 0.99 + 0.01 * var.detach(), rtol=1e-4)
    # backward
    dzb = _rand((M, ld), dt, seed=3)
    dref_in = dzb[:, c0:c0 + Cc].float().cpu()
    zr.backward(dref_in)
    dbeta, acc = torch.zeros(Cc, device="cuda"), torch.zeros(2 * Cc, dtype=torch.int64, device="cuda")
    _lib.check(lib.fn_bn_relu_train_bwd(ptr(dzb, c0), ld, ptr(ybuf, c0), ld, M, Cc, ptr(beta), ptr(sc), ptr(sh), ptr(dbeta), ptr(acc), Cc, 1, 0,
                                        0, 1, dt, stream()))
    torch.cuda.synchronize()
    assert rel_err(dzb[:, c0:c0 + Cc], yr.grad) < (8e-3 if dt == BF else 1e-3)
    dbeta_ref = (dref_in * (zr.detach() > 0)).sum(0)
    assert torch.allclose(dbeta.cpu(), dbeta_ref, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("dt", [BF, HF])
def test_maxpool_fwd_bwd(lib, dt):
    N, H, W, Cc = 2, 17, 17, 40
    xb = _rand((N, H, W, 64), dt, seed=5)
    xb[0, :6, :6, 8:24] = 0.0          # plateau of ties (post-ReLU zeros): gradient goes to the FIRST maximum
    x = xb[..., 8:8 + Cc]
    OH = OW = 8
    yb = torch.zeros(N, OH, OW, 48, dtype=lp_dtype(dt), device="cuda")
    amax = torch.zeros(N, OH, OW, Cc, dtype=torch.uint8, device="cuda")
    _lib.check(lib.fn_maxpool3x3s2_fwd(ptr(xb, 8), 64, ptr(yb, 8), 48, N, H, W, Cc, ptr(amax), dt, stream()))
    xr = x.float().cpu().permute(0, 3, 1, 2).clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2)
    assert torch.equal(yb[..., 8:8 + Cc].float().cpu(), yr.detach().permute(0, 2, 3, 1))
    dy = _rand((N, OH, OW, 48), dt, seed=6)
    yr.backward(dy[..., 8:8 + Cc].float().cpu().permute(0, 3, 1, 2))
    dx = torch.zeros(N, H, W, 64, dtype=lp_dtype(dt), device="cuda")
    _lib.check(lib.fn_maxpool3x3s2_bwd(ptr(xb, 8), 64, ptr(dy, 8), 48, ptr(dx, 8), 64, N, H, W, Cc, None, 0, dt, stream()))
    torch.cuda.synchronize()
    assert rel_err(dx[..., 8:8 + Cc], xr.grad.permute(0, 2, 3, 1)) < (4e-3 if dt == BF else 5e-4)
    dx2 = torch.zeros_like(dx)      # argmax form: identical result, no window recomputation
    _lib.check(lib.fn_maxpool3x3s2_bwd(None, 64, ptr(dy, 8), 48, ptr(dx2, 8), 64, N, H, W, Cc, ptr(amax), 0, dt, stream()))
    torch.cuda.synchronize()
    assert torch.equal(dx2, dx)
    _lib.check(lib.fn_maxpool3x3s2_bwd(ptr(xb, 8), 64, ptr(dy, 8), 48, ptr(dx, 8), 64, N, H, W, Cc, None, 1, dt, stream()))
    torch.cuda.synchronize()
    assert rel_err(dx[..., 8:8 + Cc], 2 * xr.grad.permute(0, 2, 3, 1)) < 1e-2


@pytest.mark.parametrize("dt", [BF, HF])
def test_avgpool_and_residual_bwd(lib, dt):
    N, Cc = 5, 1792
    x = _rand((N, 3, 3, Cc), dt, seed=7)
    y = torch.zeros(N, Cc, dtype=lp_dtype(dt), device="cuda")
    _lib.check(lib.fn_avgpool_fwd(ptr(x), ptr(y), N, 9, Cc, dt, stream()))
    assert rel_err(y, x.float().mean(dim=(1, 2))) < (4e-3 if dt == BF else 5e-4)
    dy = _rand((N, Cc), dt, seed=8)
    dx = torch.zeros_like(x)
    _lib.check(lib.fn_avgpool_bwd(ptr(dy), ptr(dx), N, 9, Cc, dt, stream()))
    assert rel_err(dx, (dy.float() / 9).view(N, 1, 1, Cc).expand(N, 3, 3, Cc)) < (4e-3 if dt == BF else 5e-4)
    # residual backward: out = relu(trunk + s*up)
    M = 300
    out = _rand((M, 256), dt, seed=9).clamp(min=0)
    dout = _rand((M, 256), dt, seed=10)
    dtrunk = torch.zeros_like(dout)
    dup = torch.zeros_like(dout)
    dbias = torch.zeros(256, dtype=torch.int64, device="cuda")        # fixed point, FN_ACC_GRAD_BITS
    _lib.check(lib.fn_residual_bwd(ptr(dout), ptr(out), ptr(dtrunk), ptr(dup), ptr(dbias), M, 256, 0.17, 1, 0, dt, stream()))
    torch.cuda.synchronize()
    dpre = dout.float() * (out.float() > 0)
    assert torch.equal(dtrunk.float(), dpre)
    assert rel_err(dup, 0.17 * dpre) < (4e-3 if dt == BF else 5e-4)
    assert torch.allclose(from_acc(dbias, ACC_GRAD_BITS).cpu(), (0.17 * dpre).sum(0).cpu(), rtol=2e-3, atol=2e-3)
    first = dbias.clone()                                    # integer accumulation: the same bits every run
    for _ in range(3):
        dbias.zero_()
        _lib.check(lib.fn_residual_bwd(ptr(dout), ptr(out), ptr(dtrunk), ptr(dup), ptr(dbias), M, 256, 0.17, 1, 0, dt, stream()))
        torch.cuda.synchronize()
        assert torch.equal(dbias, first)


def test_head_bn_and_l2norm(lib):
    N, E = 9, 128
    y = _rand((N, E), seed=11, scale=3.0) + 1.5
    beta = _rand((E,), seed=12, scale=0.2)
    out = torch.zeros(N, E, device="cuda")
    mm, mv = torch.zeros(E, device="cuda"), torch.ones(E, device="cuda")
    sm, sr = torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda")
    _lib.check(lib.fn_head_bn_fwd(ptr(y), ptr(out), N, E, ptr(beta), ptr(mm), ptr(mv), ptr(sm), ptr(sr), 1, 0.99, 1e-3, stream()))
    yr = y.cpu().clone().requires_grad_(True)
    o = (yr - yr.mean(0)) * torch.rsqrt(yr.var(0, unbiased=False) + 1e-3) + beta.cpu()
    assert torch.allclose(out.cpu(), o.detach(), atol=1e-5, rtol=1e-5)
    on = torch.zeros_like(out)
    _lib.check(lib.fn_l2norm_fwd(ptr(out), ptr(on), N, E, 1e-10, stream()))
    onr = fo.l2_normalize(o)
    assert torch.allclose(on.cpu(), onr.detach(), atol=1e-6)
    g = _rand((N, E), seed=13)
    onr.backward(g.cpu())
    dx = torch.zeros_like(out)
    _lib.check(lib.fn_l2norm_bwd(ptr(out), ptr(g), ptr(dx), N, E, 1e-10, stream()))
    dbeta = torch.zeros(E, device="cuda")
    dy = torch.zeros(N, E, dtype=torch.float16, device="cuda")
    _lib.check(lib.fn_head_bn_bwd(ptr(dx), ptr(y), ptr(sm), ptr(sr), ptr(dbeta), ptr(dy), N, E, HF, stream()))
    torch.cuda.synchronize()
    assert rel_err(dy, yr.grad) < 1e-3
    # eval mode uses the moving statistics
    _lib.check(lib.fn_head_bn_fwd(ptr(y), ptr(out), N, E, ptr(beta), ptr(mm), ptr(mv), None, None, 0, 0.99, 1e-3, stream()))
    assert torch.allclose(out.cpu(), (y.cpu() - mm.cpu()) * torch.rsqrt(mv.cpu() + 1e-3) + beta.cpu(), atol=1e-5)


def test_triplet_loss_and_softmax(lib):
    T, E = 7, 128
    emb = fo.l2_normalize(_rand((3 * T, E), seed=14).cpu()).cuda().contiguous()
    demb, loss = torch.zeros_like(emb), torch.zeros(4, device="cuda")      # loss: fp32[4], word 0 = the loss
    _lib.check(lib.fn_triplet_loss_fwd_bwd(ptr(emb), ptr(demb), ptr(loss), T, E, 0.2, stream()))
    er = emb.cpu().clone().requires_grad_(True)
    lr = fo.triplet_loss(er, 0.2)
    lr.backward()
    assert abs(loss[0].item() - lr.item()) < 1e-6
    assert torch.allclose(demb.cpu(), er.grad, atol=1e-6)
    # softmax cross-entropy, ragged class count with padded columns
    N, Cr, Cp = 6, 37, 40
    logits = _rand((N, Cp), seed=15, scale=3.0)
    labels = torch.tensor([0, 36, 5, 5, 17, 3], dtype=torch.int32, device="cuda")
    dl = torch.ones(N, Cp, dtype=torch.bfloat16, device="cuda")
    dbias = torch.zeros(Cp, dtype=torch.int64, device="cuda")             # fixed point, FN_ACC_GRAD_BITS
    _lib.check(lib.fn_softmax_xent_fwd_bwd(ptr(logits), Cp, ptr(labels), ptr(loss), ptr(dl), Cp, ptr(dbias), N, Cr, 1.0 / N, BF, stream()))
    lg = logits[:, :Cr].cpu().clone().requires_grad_(True)
    ref = fo.softmax_cross_entropy(lg, labels.cpu())
    ref.backward()
    assert abs(loss[0].item() - ref.item()) < 1e-5
    assert rel_err(dl[:, :Cr], lg.grad) < 5e-3 and float(dl[:, Cr:].abs().max()) == 0
    assert torch.allclose(from_acc(dbias[:Cr], ACC_GRAD_BITS).cpu(), lg.grad.sum(0), atol=1e-5)
    # an out-of-range label: the loss is NaN (TF's GPU kernel), reported through the flag word of the loss buffer
    bad = labels.clone(); bad[2] = Cr
    _lib.check(lib.fn_softmax_xent_fwd_bwd(ptr(logits), Cp, ptr(bad), ptr(loss), ptr(dl), Cp, None, N, Cr, 1.0 / N, BF, stream()))
    assert math.isnan(loss[0].item())


def test_pairwise_and_select_triplets(lib):
    from facenet_amd.statistics import pairwise_similarities
    from facenet_amd.triplet import select_triplets, squared_distances
    rng = np.random.default_rng(0)
    xa = rng.normal(size=(23, 128)).astype(np.float32)
    xa /= np.linalg.norm(xa, axis=1, keepdims=True)
    xb = rng.normal(size=(11, 128)).astype(np.float32)
    xb /= np.linalg.norm(xb, axis=1, keepdims=True)
    for metric in (0, 1):
        assert np.allclose(pairwise_similarities(xa, xb, metric), fo.pairwise_similarities(xa, xb, metric), atol=2e-6 if metric == 0 else 2e-4)
        assert np.allclose(pairwise_similarities(xa, None, metric), fo.pairwise_similarities(xa, None, metric), atol=2e-6 if metric == 0 else 2e-4)
    assert pairwise_similarities(xa[:1], None).shape == (0,)             # statistics.py:32-36: empty upper triangle
    with pytest.raises(ValueError):
        pairwise_similarities(xa * 3.0, xa)                                # :40-42 not normalised (diagonal = 3)
    with pytest.raises(ValueError):
        pairwise_similarities(xa, xb, metric=2)                            # :55
    # selection: exact index match with the oracle given the same distances and hash stream
    P, K = 12, 4
    labels = np.repeat(np.arange(P), K)
    emb = rng.normal(size=(P * K, 64)).astype(np.float32) * 0.05 + rng.normal(size=(P, 1, 64)).astype(np.float32).repeat(K, 1).reshape(P * K, 64) * 0.08
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    dist = squared_distances(torch.from_numpy(emb).cuda())
    dref = fo.squared_distance_matrix(emb)
    assert np.array_equal(dist.cpu().numpy(), dref)            # the oracle sums in the device's order: same bits (E = 64: one register per lane)
    for semi in (False, True):
        for seed in (0, 123):
            trip, info = select_triplets(dist, labels, 0.2, 30, seed=seed, semi_hard=semi)
            ref = fo.select_triplets(dist.cpu().numpy(), labels, 0.2, 30, seed, semi_hard=semi)
            assert np.array_equal(trip.cpu().numpy(), ref), (semi, seed)
            assert info["pairs"] == P * K * (K - 1) // 2
    with pytest.raises(ValueError):
        select_triplets(dist, labels, 0.2, 10_000)


def test_adam_keras_and_packs(lib):
    n, n_decay, n_lp = 4096, 3000, 2048
    g0 = torch.Generator().manual_seed(1)
    w = torch.randn(n, generator=g0).cuda()
    params = {"w": w.cpu().clone()}
    opt = fo.AdamKeras(["w"], params, lr=0.05)
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    wlp = torch.zeros(n_lp, dtype=torch.bfloat16, device="cuda")
    hyper = torch.tensor([0.05, 1.0, 1.0, 0.5, 0.0, 0.0, 0.0, 0.0], device="cuda")   # word 4: int32 step count
    for step in range(3):
        g = torch.randn(n, generator=g0)
        gg = g * 0.5
        gg[:n_decay] += 2 * 5e-4 * params["w"][:n_decay]
        opt.step(params, {"w": gg})
        gd = g.cuda()
        _lib.check(lib.fn_adam_tick(ptr(hyper), 0.9, 0.999, stream()))
        _lib.check(lib.fn_adam_keras(ptr(w), ptr(gd), ptr(m), ptr(v), ptr(wlp), n_lp, n, n_decay, ptr(hyper), 0.9, 0.999, 0.1, 5e-4, BF, stream()))
    torch.cuda.synchronize()
    assert torch.allclose(w.cpu(), params["w"], atol=2e-6, rtol=1e-5)
    assert torch.equal(wlp.cpu(), params["w"][:n_lp].to(torch.bfloat16)) or rel_err(wlp, params["w"][:n_lp]) < 4e-3
