"""Full-size checks (BASELINE.json configs[1] batch 90; configs[3] 10 575 classes) through size-independent properties:
the CPU oracle would need minutes at these sizes, the identities below need none.

Adjoint identities of a convolution y = conv(x, w):   <dy, conv(x, w)> == <dgrad(dy, w), x> == <wgrad(x, dy), w>
(they tie fwd, dgrad and wgrad together at the real layer shapes, stride-2 parity classes and split-K included),
linearity of the forward kernel, unit-norm embeddings, and replica-free invariants of one full training step."""
import ctypes as C

import numpy as np
import pytest
import torch

from facenet_amd import _lib
from tests.util import conv_desc, lp_dtype, ptr, stream

pytestmark = pytest.mark.gpu
BF = _lib.FN_BF16

LAYERS = [  # N, H, W, Cin, Cout, kh, kw, stride, ph, pw   (batch-90 shapes from the SURVEY.md shape table)
    (90, 79, 79, 32, 32, 3, 3, 1, 0, 0),      # Conv2d_2a
    (90, 35, 35, 192, 256, 3, 3, 2, 0, 0),    # Conv2d_4b (stride 2)
    (90, 17, 17, 256, 32, 1, 1, 1, 0, 0),     # block35 1x1
    (90, 17, 17, 256, 384, 3, 3, 2, 0, 0),    # reduction_a stride 2
    (90, 8, 8, 128, 128, 1, 7, 1, 0, 3),      # block17 1x7
    (90, 8, 8, 256, 896, 1, 1, 1, 0, 0),      # block17 up
    (90, 3, 3, 1792, 192, 1, 1, 1, 0, 0),     # block8 1x1
    (90, 3, 3, 192, 192, 3, 1, 1, 1, 0),      # block8 3x1
]


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, device="cuda", generator=g) * scale).to(torch.bfloat16)


@pytest.mark.parametrize("case", LAYERS)
def test_conv_adjoint_identities_at_full_size(lib, case):
    N, H, W, Cin, Cout, kh, kw, s, ph, pw = case
    d = conv_desc(N, H, W, Cin, Cout, kh, kw, s, ph, pw, BF)
    x = _rand((N, H, W, Cin), 1)
    w = _rand((Cout, kh, kw, Cin), 2, 0.05)
    dy = _rand((N, d.OH, d.OW, Cout), 3)
    # forward (fp32 output so the inner products are limited by the operands' rounding only)
    y = torch.zeros(N, d.OH, d.OW, Cout, dtype=torch.float32, device="cuda")
    d.x, d.w, d.y, d.out_f32 = ptr(x), ptr(w), ptr(y), 1
    _lib.check(lib.fn_conv2d_fwd(C.byref(d), stream()))
    # dgrad (fp32 output)
    wt = torch.zeros_like(w).view(-1)
    table = torch.tensor([[0, Cout, kh * kw * Cin, kh * kw, Cin, -1, -1, 0]], dtype=torch.int32, device="cuda")
    _lib.check(lib.fn_pack_transpose(ptr(w), ptr(wt), ptr(table), 1, w.numel(), BF, stream()))
    dx = torch.zeros(N, H, W, Cin, dtype=torch.float32, device="cuda")
    g = conv_desc(N, H, W, Cin, Cout, kh, kw, s, ph, pw, BF)
    g.y, g.w, g.dx, g.out_f32 = ptr(dy), ptr(wt), ptr(dx), 1
    _lib.check(lib.fn_conv2d_dgrad(C.byref(g), stream()))
    # wgrad
    dw = torch.zeros(Cout, kh, kw, Cin, dtype=torch.float32, device="cuda")
    q = conv_desc(N, H, W, Cin, Cout, kh, kw, s, ph, pw, BF)
    q.x, q.y, q.dw = ptr(x), ptr(dy), ptr(dw)
    _lib.check(lib.fn_conv2d_wgrad(C.byref(q), stream()))
    torch.cuda.synchronize()
    a = (dy.double() * y.double()).sum().item()
    b = (dx.double() * x.double()).sum().item()
    c = (dw.double() * w.double()).sum().item()
    scale = (dy.double().norm() * y.double().norm()).item()
    assert abs(a - b) < 2e-5 * scale and abs(a - c) < 2e-5 * scale, (a, b, c, scale)
    assert torch.isfinite(y).all() and y.abs().max() > 0


def test_conv_linearity_at_full_size(lib):
    N, H, W, Cin, Cout = 90, 17, 17, 192, 192
    d = conv_desc(N, H, W, Cin, Cout, 3, 3, 1, 1, 1, BF)
    x1, x2 = _rand((N, H, W, Cin), 4), _rand((N, H, W, Cin), 5)
    x12 = (x1.float() + x2.float()).to(torch.bfloat16)
    x12f = x12.float()                                  # the exactly representable sum
    w = _rand((Cout, 3, 3, Cin), 6, 0.05)
    outs = []
    for xin in (x1, x2, x12):
        y = torch.zeros(N, H, W, Cout, dtype=torch.float32, device="cuda")
        d.x, d.w, d.y, d.out_f32 = ptr(xin), ptr(w), ptr(y), 1
        _lib.check(lib.fn_conv2d_fwd(C.byref(d), stream()))
        outs.append(y)
    torch.cuda.synchronize()
    # conv(x1) + conv(x2) - conv(round(x1+x2)) == conv(x1 + x2 - round(x1+x2)): small, and exactly linear in fp32 accumulate
    resid = x1.float() + x2.float() - x12f
    yr = torch.nn.functional.conv2d(resid.permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    err = (outs[0] + outs[1] - outs[2] - yr).norm() / outs[2].norm()
    assert err < 1e-5, err


def test_full_batch_step_invariants():
    """One configs[1] step (batch 90 = 30 triplets, bf16): eval embeddings are unit-norm, the hinge loss is bounded,
    gradients are finite, beta gradients of the layers in front of another BatchNorm vanish in the mean... and the
    Keras-Adam update is bounded by lr_t (|m|/(sqrt(v)+eps) <= ~sqrt(1-b2)/(1-b1) scaled)."""
    from facenet_amd.engine import Network
    from facenet_amd.train import Trainer
    from tests.util_data import structured_images
    net = Network(embedding_size=128, device="cuda:0")
    x = torch.from_numpy(structured_images(90, seed=11))
    plan = net.plan(90, training=False)
    plan.images.copy_(x)
    net.refresh_folded(net.stream())
    plan.run_forward()
    emb = plan.embedding.buf.act.view(90, 128).float()
    embn = torch.nn.functional.normalize(emb, dim=1)
    assert torch.isfinite(emb).all() and torch.allclose(embn.norm(dim=1), torch.ones(90, device="cuda"), atol=1e-5)
    p0 = net.P.clone()
    tr = Trainer(net, batch=90, loss="triplet", alpha=0.2, lr=0.05)
    tr.set_images(x)
    tr.step()
    torch.cuda.synchronize()
    loss = tr.loss_value()
    assert 0.0 <= loss <= 4.2                      # |a-p|^2 - |a-n|^2 + alpha with unit vectors
    assert torch.isfinite(tr.G).all() and tr.G.abs().max() > 0
    lr_t = 0.05 * np.sqrt(1 - 0.999) / (1 - 0.9)
    assert (net.P - p0).abs().max().item() <= lr_t * 1.0001 / 0.0316 * 0.1 + 1e-6   # |m|/(sqrt(v)+eps) <= 0.1|g|/(0.0316|g|)
    # the embedding of the training forward is batch-normalised: zero mean, ~unit variance per feature
    e = tr.emb.float()
    assert e.mean(0).abs().max().item() < 2e-2 and abs(e.var(0, unbiased=False).mean().item() - 1.0) < 5e-2


def test_softmax_config4_classifier_10575_classes():
    """BASELINE.json configs[3]: E=512, 10 575 classes (ragged: padded to 10 576 columns).  Loss and classifier gradients
    against the CPU oracle on the same embeddings; the backbone is shared with the other tests."""
    from facenet_amd.engine import Network
    from facenet_amd.train import Trainer
    from oracle import facenet_oracle as fo
    from tests.util_data import structured_images
    E, N, Cc = 512, 8, 10575
    net = Network(embedding_size=E, device="cuda:0", nrof_classes=Cc, train_dtype=torch.float16)
    tr = Trainer(net, batch=N, loss="softmax", l2=0.0)
    labels = np.random.default_rng(9).integers(0, Cc, N)
    tr.set_images(torch.from_numpy(structured_images(N, seed=12)), torch.from_numpy(labels))
    st = net.stream()
    # (the bias gradient is accumulated in fixed point; `grad_finalize`, the last launch of the backward list, moves it into G)
    for ops in (tr.pre_ops, tr.plan.fwd, tr.loss_ops, [op for op in tr.plan.bwd if op.name == "grad_finalize"]):
        tr.plan.run_ops(ops, st)
    torch.cuda.synchronize()
    params = net.export_keras_params()
    emb = tr.emb.float().cpu()
    wk = params["classifier/logits/kernel"].clone().requires_grad_(True)
    bk = params["classifier/logits/bias"].clone().requires_grad_(True)
    assert wk.shape == (E, Cc)
    ref = fo.softmax_cross_entropy(emb.to(torch.float16).float() @ wk.to(torch.float16).float() + bk, torch.as_tensor(labels))
    ref.backward()
    assert abs(tr.loss_value() - ref.item()) < 2e-3 * ref.item()
    g = net.export_keras_grads(tr.G)
    assert (g["classifier/logits/kernel"] - wk.grad).norm() / wk.grad.norm() < 2e-2
    assert (g["classifier/logits/bias"] - bk.grad).norm() / bk.grad.norm() < 2e-2
    assert float(tr.logits[:, Cc:].abs().max()) == 0 and float(tr.dlogits[:, Cc:].float().abs().max()) == 0   # padded column
