"""Training-loop plumbing on the GPU: the assembled TripletMiner (mining forward -> distances -> selection -> gather), graph
capture without side effects, label validation, checkpoints in Keras variable naming, the BN-folded inference export and a
well-conditioned whole-network gradient check.  PARITY UNPINNED by the reference: the oracle is this repo's
CPU restatement; triplet selection has no reference file at all."""
import numpy as np
import pytest
import torch

from facenet_amd.engine import BN_EPS, Network
from facenet_amd.train import GraphRunner, Trainer, TripletMiner
from facenet_amd.schedule import make_events
from oracle import facenet_oracle as fo
from tests.util_data import structured_images

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("graph", [False, True])
def test_triplet_miner_end_to_end(graph):
    """TripletMiner.run(): for replays k = 0, 1, 2 the selected triplets equal the oracle's selection on the DEVICE distance
    matrix with seed + k (the kernels step the seed by a call counter so that a replayed HIP graph, whose arguments are
    frozen, still draws fresh negatives), and the gathered train batch is bit-equal to pool[triplets]."""
    P, K, T, alpha, seed = 12, 4, 10, 0.2, 7
    n = P * K
    labels = np.repeat(np.arange(P), K)
    net = Network(embedding_size=128, device="cuda:0")
    params, _, _ = fo.build_params(128, seed=0)
    fo.perturb_bn_stats(params, seed=1)
    net.load_keras_params(params)
    miner = TripletMiner(net, n, labels, T, alpha=alpha, seed=seed)
    train_images = torch.zeros(3 * T, 160, 160, 3, dtype=torch.uint8, device="cuda:0")
    miner.build(train_images)
    pools = [torch.from_numpy(structured_images(n, seed=30 + k)) for k in range(3)]
    runner = None
    if graph:
        miner.plan.images.copy_(pools[0])
        miner.run()                                   # warm-up outside the capture: advances the call counter
        torch.cuda.synchronize()
        miner.info.zero_()
        ev = make_events(miner.sched)
        runner = GraphRunner(net.device).capture(lambda: miner.run(ev))
        miner.info.zero_()                            # a capture records, it does not execute
    for k in range(3):
        miner.plan.images.copy_(pools[k])
        runner.replay() if runner is not None else miner.run()
        torch.cuda.synchronize()
        # mining embeddings are the inference path's (f16 storage): unit rows, close to the oracle
        emb = miner.embn.cpu()
        assert torch.allclose(emb.norm(dim=1), torch.ones(n), atol=1e-5)
        ref = fo.Oracle(params).forward(pools[k].numpy(), training=False)
        assert (emb - ref).norm(dim=1).max().item() <= 1e-3
        dist = miner.dist.cpu().numpy()
        assert np.array_equal(dist, dist.T)           # the selection kernel reads columns instead of rows
        want = fo.select_triplets(dist, labels, alpha, T, seed=seed + k)
        got = miner.triplets.cpu().numpy()
        assert np.array_equal(got, want), (k, got, want)
        info = miner.info[:4].cpu().numpy()
        assert info[0] == P * K * (K - 1) // 2 and info[2] == 0 and info[3] == k + 1
        gathered = train_images.cpu().numpy()
        assert np.array_equal(gathered, pools[k].numpy()[got.reshape(-1)])     # bytes, bit for bit


def test_miner_rejects_pools_that_cannot_fill_the_batch():
    net = Network(embedding_size=128, device="cuda:0")
    with pytest.raises(ValueError):
        TripletMiner(net, 8, [0] * 8, 4)                       # one identity: no negatives
    with pytest.raises(ValueError):
        TripletMiner(net, 8, list(range(8)), 4)                # no anchor-positive pair
    with pytest.raises(ValueError):
        TripletMiner(net, 8, [0, 0, 1, 1, 2, 3, 4, 5], 3)      # 2 pairs < 3 triplets


def test_capture_leaves_training_state_untouched():
    """Trainer.capture() warms up with a real step; parameters, moving statistics, Adam moments and Adam's t must come
    back, so that capture() + n steps is n steps (the reference's fit() has no uncounted step)."""
    net = Network(embedding_size=128, device="cuda:0", train_dtype=torch.float16)
    tr = Trainer(net, batch=6, loss="triplet", alpha=0.2, lr=0.01)
    x = structured_images(6, seed=2)
    x[2], x[5] = x[1], x[4]                                    # negatives = positives: every triplet is active
    tr.set_images(torch.from_numpy(x))
    before = [t.clone() for t in (net.P, net.S_mean, net.S_var, tr.M, tr.V, tr.hyper, net.W_train, net.Wt_train)]
    tr.capture()
    torch.cuda.synchronize()
    for b, t in zip(before, (net.P, net.S_mean, net.S_var, tr.M, tr.V, tr.hyper, net.W_train, net.Wt_train)):
        assert torch.equal(b, t)
    tr.step()
    torch.cuda.synchronize()
    h = tr.hyper.cpu().numpy()
    assert abs(h[1] - 0.9) < 1e-6 and abs(h[2] - 0.999) < 1e-6 and tr.iterations == 1   # exactly one optimiser step: t = 1
    assert float((net.P - before[0]).abs().max()) > 0


@pytest.mark.parametrize("kind", ["triplet", "softmax"])
def test_training_steps_are_reproducible_bit_for_bit(kind):
    """Two trainers from the same weights taking the same three steps end with identical bits: losses, gradients, parameters,
    moving statistics, Adam moments (verdict r02: BN sums and dW went through fp32 atomics and every run differed)."""
    ncls = 23 if kind == "softmax" else None
    xs = structured_images(9, seed=11)
    xs[2::3] = xs[1::3]                    # negatives = positives: every triplet stays active (loss = alpha) through all three steps
    x = torch.from_numpy(xs)
    labels = torch.tensor([i % 23 for i in range(9)]) if kind == "softmax" else None
    runs = []
    for _ in range(2):
        net = Network(embedding_size=128, device="cuda:0", nrof_classes=ncls, seed=3)
        tr = Trainer(net, batch=9, loss=kind, alpha=0.2, lr=0.05)
        tr.set_images(x, labels)
        losses = []
        for _ in range(3):
            tr.step()
            torch.cuda.synchronize()
            losses.append(tr.loss_value())
        runs.append((losses, tr.G.clone(), net.P.clone(), net.S_mean.clone(), net.S_var.clone(), tr.M.clone(), tr.V.clone()))
    assert runs[0][0] == runs[1][0] and float(runs[0][1].abs().max()) > 0
    for a, b in zip(runs[0][1:], runs[1][1:]):
        assert torch.equal(a, b)


def test_capture_refuses_more_than_two_streams():
    """Captured schedules span at most 2 streams (hipStreamEndCapture faults beyond that under torch.cuda.graph: DESIGN.md 5)."""
    net = Network(embedding_size=128, device="cuda:0")
    tr = Trainer(net, batch=3, loss="triplet", n_streams=3)
    with pytest.raises(ValueError):
        tr.capture()


def test_softmax_labels_are_validated():
    net = Network(embedding_size=128, device="cuda:0", nrof_classes=37)
    tr = Trainer(net, batch=4, loss="softmax")
    x = torch.from_numpy(structured_images(4, seed=1))
    tr.set_images(x, torch.tensor([0, 36, 5, 7]))
    for bad in ([0, 37, 5, 7], [-1, 3, 5, 7], [1, 2, 3]):
        with pytest.raises(ValueError):
            tr.set_images(x, torch.tensor(bad))
    tri = Trainer(Network(embedding_size=128, device="cuda:0"), batch=3, loss="triplet")
    with pytest.raises(ValueError):
        tri.set_images(x[:3], torch.tensor([0, 1, 2]))


def test_checkpoint_round_trip_in_keras_naming(tmp_path):
    """apps/train_softmax.py:68-78,105: weights are saved and restored through Keras.  The checkpoint holds the reference
    model's variables under their Keras names in model.weights order, plus the Adam slots, iteration count and epoch."""
    from facenet_amd import keras_names
    net = Network(embedding_size=128, device="cuda:0", nrof_classes=11, train_dtype=torch.float16)
    tr = Trainer(net, batch=4, loss="softmax", lr=0.01)
    tr.set_images(torch.from_numpy(structured_images(4, seed=1)), torch.tensor([0, 3, 10, 7]))
    for _ in range(2):
        tr.step()
    torch.cuda.synchronize()
    path = tmp_path / "ckpt.npz"
    tr.save_checkpoint(path, epoch=5)
    with np.load(path) as z:
        names = list(z.files)
        model_names = [k for k in names if not k.startswith("Adam/") and k != "epoch"]
        assert model_names == [k for k, _ in keras_names.keras_variable_table(net.layers)]
        assert model_names[0] == "inception_resnet_v1/conv2d/Conv2d_1a_3x3/kernel:0"
        assert z[model_names[0]].shape == (3, 3, 3, 32)                                     # HWIO, un-padded
        assert "inception_resnet_v1/features/batch_normalization_111/moving_variance:0" in model_names
        assert z["sequential_52/logits/kernel:0"].shape == (128, 11)                         # Dense: [in, out]
        assert int(z["Adam/iter:0"]) == 2 and int(z["epoch"]) == 5
        assert "Adam/inception_resnet_v1/block8_5/Conv2d_1x1/bias/m:0" in names
    net2 = Network(embedding_size=128, device="cuda:0", nrof_classes=11, train_dtype=torch.float16, seed=99)
    tr2 = Trainer(net2, batch=4, loss="softmax", lr=0.5)
    assert tr2.load_checkpoint(path) == 5
    real = torch.ones(net.n_params, dtype=torch.bool)          # padded rows / channels are not variables: compare through Keras
    a, b = net.export_keras_params(), net2.export_keras_params()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    for buf, buf2 in ((tr.M, tr2.M), (tr.V, tr2.V)):
        ga, gb = net.export_keras_grads(buf), net2.export_keras_grads(buf2)
        for k in ga:
            assert torch.equal(ga[k], gb[k]), k
    assert torch.equal(tr.hyper[:3], tr2.hyper[:3]) and tr2.iterations == tr.iterations == 2
    assert torch.equal(net.W_train, net2.W_train)              # packs were refreshed from the restored masters
    # the model alone restores from the same file (InceptionResnetV1.load_weights path: names, not positions)
    net3 = Network(embedding_size=128, device="cuda:0", nrof_classes=11, seed=5)
    with np.load(path) as z:
        net3.load_keras_params({k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("Adam/") and k != "epoch"})
    assert torch.equal(net3.P, net.P)
    # Adam's t is an integer of its own: a checkpoint written after more steps than beta1^t survives in fp32 (~970) keeps it
    tr.iterations = 1100
    tr.step()
    torch.cuda.synchronize()
    from facenet_amd.train import adam_beta_powers
    assert tr.iterations == 1101 and float(tr.hyper[1]) == 0.0 and float(tr.hyper[2]) == adam_beta_powers(1101, 0.9, 0.999)[1]
    assert abs(float(tr.hyper[2]) - 0.999 ** 1101) < 1e-5           # (the betas are fp32 on the device, as Keras' hyper-parameters are)
    late = tmp_path / "late.npz"
    tr.save_checkpoint(late, epoch=1)
    with np.load(late) as z:
        assert int(z["Adam/iter:0"]) == 1101
    tr2.load_checkpoint(late)
    assert tr2.iterations == 1101 and torch.equal(tr.hyper.view(torch.int32)[:5], tr2.hyper.view(torch.int32)[:5])
    tr.step(); tr2.step()
    torch.cuda.synchronize()
    assert tr2.iterations == 1102 and torch.equal(tr.hyper.view(torch.int32)[:5], tr2.hyper.view(torch.int32)[:5])


def test_bn_folded_export_matches_tfutils_formula():
    """facenet/tfutils.py:244-250: weights * 1/sqrt(variance + epsilon), biases = -mean * scale + beta, applied to the
    un-folded Keras variables, must give what the inference kernels read (fn_fold_bn), to f16 rounding."""
    params, _, _ = fo.build_params(128, seed=0)
    fo.perturb_bn_stats(params, seed=1)
    net = Network(embedding_size=128, device="cuda:0", infer_dtype=torch.float16)
    net.load_keras_params(params)
    folded = net.export_folded_params()
    n_bn = 0
    for L in net.layers.values():
        w = params[L.name + "/kernel"]
        if L.has_bn:
            pre = "features/bn" if L.name == "features/logits" else L.name + "/bn"
            scale = 1.0 / torch.sqrt(params[pre + "/moving_variance"] + BN_EPS)
            want_w = w * scale                                           # HWIO / [in,out]: the output channel is the last axis
            want_b = -params[pre + "/moving_mean"] * scale + params[pre + "/beta"]
            n_bn += 1
        else:
            want_w, want_b = w, params[L.name + "/bias"]
        got_w, got_b = folded[L.name + "/weights"], folded[L.name + "/biases"]
        assert got_w.shape == want_w.shape
        assert torch.allclose(got_w, want_w.half().float(), rtol=0, atol=0) or \
            (got_w - want_w).abs().max() <= 2 ** -10 * want_w.abs().max(), L.name      # one f16 rounding of the product
        assert torch.allclose(got_b, want_b, rtol=1e-6, atol=1e-6), L.name
    assert n_bn == 112


TRUNC = {"block35": {"repeat": 1, "scale": 0.17, "activation": "relu"},
         "block17": {"repeat": 0, "scale": 0.10, "activation": "relu"},
         "block8_1": {"repeat": 0, "scale": 0.2, "activation": "relu"}}


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_truncated_network_gradients_at_batch_45(dt):
    """Whole-model gradient check on a shortened network: stem + 1 x Block35 + ReductionA + ReductionB + the last Block8 + head
    + the softmax classifier (apps/train_softmax.py:49-104) at batch 45, against three references: the fp32 oracle, the
    storage-rounding model (tests/quant_oracle.py) and the storage-rounding model on the device's ReLU active sets.
    Measured: the rounding model itself sits at cosine 0.993 (f16) / 0.950 (bf16) from fp32 -- 16-bit activations flip the
    sign of near-zero pre-activations, and no implementation with that storage gets closer -- so the HIP path is held to
    "as good as the rounding model" against fp32 and to cosine >= 0.999 (f16) / 0.995 (bf16) against the rounding model once the
    ReLU active sets are shared.  The block-level tests (test_gpu_blocks.py) hold the backward arithmetic itself to 2e-3 / 1e-2."""
    from tests.quant_oracle import QuantOracle, _q
    E, N, NC = 128, 45, 37
    params, trainable, _ = fo.build_params(E, seed=0, config=TRUNC, nrof_classes=NC)
    x = structured_images(N, seed=9)
    labels = np.random.default_rng(5).integers(0, NC, N)
    loss_ref, _, g32, _, emb_ref = fo.train_step_grads(params, trainable, [], x, "softmax", labels=labels, config=TRUNC)
    # storage-rounding model
    for k in trainable:
        params[k].requires_grad_(True)
        params[k].grad = None
    qo = QuantOracle(params, dt, config=TRUNC)
    emb_q = qo.forward(x, training=True)
    logits = _q(emb_q, dt) @ _q(params["classifier/logits/kernel"], dt) + params["classifier/logits/bias"]
    fo.softmax_cross_entropy(logits, torch.as_tensor(labels)).backward()
    gq = {k: params[k].grad.detach().clone() for k in trainable}
    for k in trainable:
        params[k].requires_grad_(False)
        params[k].grad = None
    net = Network(embedding_size=E, device="cuda:0", train_dtype=dt, config=TRUNC, nrof_classes=NC)
    net.load_keras_params(params)
    tr = Trainer(net, batch=N, loss="softmax", l2=0.0)
    tr.set_images(torch.from_numpy(x), torch.from_numpy(labels))
    st = net.stream()
    tr._zero()
    for ops in (tr.plan.fwd, tr.loss_ops, tr.plan.bwd):
        tr.plan.run_ops(ops, st)
    torch.cuda.synchronize()
    mine = net.export_keras_grads(tr.G)
    keys = [k for k in trainable if g32[k].norm() > 1e-6]
    flat = lambda d: torch.cat([d[k].reshape(-1).double() for k in keys])
    cosine = lambda a, b: torch.nn.functional.cosine_similarity(flat(a), flat(b), dim=0).item()
    relerr = lambda a, b: (flat(a) - flat(b)).norm().item() / flat(b).norm().item()
    # third reference: the rounding model evaluated on the DEVICE's ReLU active sets (quant_oracle._ReluWithMask): what is left
    # between it and the HIP path is arithmetic, not the sign of near-zero pre-activations
    masks = {}
    for r in tr.plan.recs:
        if r.kind == "conv" and r.extra.get("kind") == "bn":
            masks[r.layer.name] = (r.y.buf.act[..., r.y.c0:r.y.c0 + r.y.C] > 0).float().cpu().permute(0, 3, 1, 2).contiguous()
        elif r.kind == "conv" and r.extra.get("kind") == "resid" and r.extra["relu"]:
            masks[r.layer.name[:-3]] = (r.y.buf.act > 0).float().cpu().permute(0, 3, 1, 2).contiguous()
    for k in trainable:
        params[k].requires_grad_(True)
        params[k].grad = None
    qm = QuantOracle(params, dt, masks=masks, config=TRUNC)
    logits = _q(qm.forward(x, training=True), dt) @ _q(params["classifier/logits/kernel"], dt) + params["classifier/logits/bias"]
    fo.softmax_cross_entropy(logits, torch.as_tensor(labels)).backward()
    gm = {k: params[k].grad.detach().clone() for k in trainable}
    for k in trainable:
        params[k].requires_grad_(False)
        params[k].grad = None
    cos, cos_q, cos_m = cosine(mine, g32), cosine(gq, g32), cosine(mine, gm)
    rel, rel_q, rel_m = relerr(mine, g32), relerr(gq, g32), relerr(mine, gm)
    print(f"{dt}: loss {tr.loss_value():.5f} (fp32 {loss_ref:.5f}); cosine HIP-fp32 {cos:.5f} (rounding model-fp32 {cos_q:.5f}), HIP-rounding "
          f"model on device masks {cos_m:.6f}; rel err {rel:.4f} ({rel_q:.4f}), {rel_m:.4f}")
    f16 = dt == torch.float16
    assert abs(tr.loss_value() - loss_ref) < (2e-3 if f16 else 2e-2)
    # against fp32 nobody storing activations in 16 bits can do better than the rounding model: ReLU sign flips of near-zero
    # pre-activations put it at cosine 0.993 (f16) / 0.95 (bf16) even at this batch.  The HIP path must be as good ...
    assert cos >= cos_q - 0.003 and rel <= 1.1 * rel_q + 0.01
    assert cos >= (0.99 if f16 else 0.94)
    # ... and, with the ReLU active sets shared, agree with the rounding model much more closely (measured 0.9995 / 0.998; what
    # remains are the max-pool selections, which flip the same way on near-ties and are not shared, and the arithmetic)
    assert cos_m >= (0.999 if f16 else 0.995) and rel_m <= (0.05 if f16 else 0.10)
    assert cos_m > cos + (0.004 if f16 else 0.03)
