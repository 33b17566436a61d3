"""End-to-end parity of the HIP path against the CPU fp32 oracle (oracle/facenet_oracle.py).

PARITY UNPINNED by the reference (it ships no vectors): the oracle is this repo's restatement."""
import numpy as np
import pytest
import torch

from facenet_amd.engine import Network
from facenet_amd.train import Trainer
from oracle import facenet_oracle as fo

pytestmark = pytest.mark.gpu


def _images(n, seed=0):
    return np.random.default_rng(seed).integers(0, 256, (n, 160, 160, 3), dtype=np.uint8)


@pytest.fixture(scope="module")
def net128():
    net = Network(embedding_size=128, device="cuda:0")
    return net


def test_variable_counts(net128):
    assert net128.count_variables() == (22808144, 22779312)      # SURVEY.md shape table, E=128


def test_init_matches_oracle_declaration_order(net128):
    params, _, _ = fo.build_params(128, seed=0)
    mine = net128.init_keras_params(0)
    assert list(mine.keys()) == list(params.keys())
    for k in params:
        assert torch.equal(mine[k], params[k]), k


@pytest.mark.parametrize("variant", ["fresh", "perturbed"])
@pytest.mark.parametrize("E", [128, 512])
def test_forward_embeddings_c1(variant, E):
    """BASELINE.json configs[0] / SURVEY.md 8d C1: 16 random uint8 images, unit-norm embeddings, max row L2 error
    <= 1e-3 (f16 storage, fp32 accumulate).  bf16 storage is reported against its own looser bound (2e-2)."""
    params, _, _ = fo.build_params(E, seed=0)
    if variant == "perturbed":
        fo.perturb_bn_stats(params, seed=1)
    x = _images(16)
    ref = fo.Oracle(params).forward(x, training=False)
    for dt, tol in ((torch.float16, 1e-3), (torch.bfloat16, 2e-2)):
        net = Network(embedding_size=E, device="cuda:0", infer_dtype=dt)
        net.load_keras_params(params)
        plan = net.plan(16, training=False)
        plan.images.copy_(torch.from_numpy(x))
        plan.run_forward()
        torch.cuda.synchronize()
        emb = fo.l2_normalize(plan.embedding.buf.act.view(16, E).float().cpu())
        err = (emb - ref).norm(dim=1).max().item()
        print(f"C1 {variant} E={E} {dt}: max row L2 err {err:.3e}")
        assert err <= tol
        assert torch.allclose(emb.norm(dim=1), torch.ones(16), atol=1e-5)


def _grad_check(net, grads_ref, G, tol_w, tol_small):
    mine = net.export_keras_grads(G)
    worst = ("", 0.0)
    for k, g in grads_ref.items():
        denom = g.norm().item()
        err = (mine[k] - g).norm().item() / (denom + 1e-12)
        if err > worst[1]:
            worst = (k, err)
        lim = tol_w if k.endswith("kernel") else tol_small
        assert err < lim or (mine[k] - g).abs().max().item() < 1e-6, f"{k}: rel err {err:.3e} (|g|={denom:.3e})"
    return worst


def test_triplet_train_step_gradients():
    """forward(training=True) + l2_normalize + triplet loss + backward vs autograd on the oracle (batch 9 = 3 triplets)."""
    E, N = 128, 9
    params, trainable, regularized = fo.build_params(E, seed=0)
    x = _images(N, seed=3)
    loss_ref, _, grads_ref, new_stats, emb_ref = fo.train_step_grads(params, trainable, [], x, "triplet", alpha=0.2)
    net = Network(embedding_size=E, device="cuda:0")
    net.load_keras_params(params)
    tr = Trainer(net, batch=N, loss="triplet", alpha=0.2, l2=0.0)
    tr.set_images(torch.from_numpy(x))
    st = net.stream()
    tr._zero()
    tr.plan.run_ops(tr.plan.fwd, st)
    tr.plan.run_ops(tr.loss_ops, st)
    tr.plan.run_ops(tr.plan.bwd, st)
    torch.cuda.synchronize()
    emb = tr.emb.float().cpu()
    rel = ((emb - emb_ref).norm() / emb_ref.norm()).item()
    print(f"train-mode embedding rel err {rel:.3e}; loss {tr.loss_value():.5f} vs {loss_ref:.5f}")
    assert rel < 3e-2
    assert abs(tr.loss_value() - loss_ref) < 2e-2 * max(1.0, abs(loss_ref))
    worst = _grad_check(net, grads_ref, tr.G, tol_w=0.12, tol_small=0.12)
    print("worst gradient", worst)
    # moving statistics follow momentum 0.99 with the biased batch variance
    mine = net.export_keras_params()
    for k in ("conv2d/Conv2d_1a_3x3/bn/moving_mean", "block17/3/tower_conv1/Conv2d_0b_1x7/bn/moving_variance", "features/bn/moving_mean"):
        assert torch.allclose(mine[k], new_stats[k], rtol=3e-2, atol=3e-3), k


def test_softmax_train_step_gradients():
    """apps/train_softmax.py:49-104 step: classifier Dense + sparse softmax cross-entropy (batch 8, 37 classes)."""
    E, N, Cc = 128, 8, 37
    params, trainable, regularized = fo.build_params(E, seed=0, nrof_classes=Cc)
    x = _images(N, seed=4)
    labels = np.random.default_rng(5).integers(0, Cc, N)
    loss_ref, _, grads_ref, _, _ = fo.train_step_grads(params, trainable, [], x, "softmax", labels=labels)
    net = Network(embedding_size=E, device="cuda:0", nrof_classes=Cc)
    net.load_keras_params(params)
    tr = Trainer(net, batch=N, loss="softmax", l2=0.0)
    tr.set_images(torch.from_numpy(x), torch.from_numpy(labels))
    st = net.stream()
    tr._zero()
    tr.plan.run_ops(tr.plan.fwd, st)
    tr.plan.run_ops(tr.loss_ops, st)
    tr.plan.run_ops(tr.plan.bwd, st)
    torch.cuda.synchronize()
    print(f"softmax loss {tr.loss_value():.5f} vs {loss_ref:.5f}")
    assert abs(tr.loss_value() - loss_ref) < 2e-2 * max(1.0, abs(loss_ref))
    _grad_check(net, grads_ref, tr.G, tol_w=0.12, tol_small=0.12)
