"""The CPU oracle against (1) the committed golden fixtures (drift guard), (2) the structural anchors the reference's
own files provide (variable counts / layer counts / MACs, SURVEY.md section 8a shape table), and (3) an independent
NumPy direct-loop convolution + BatchNorm, which shares no code with torch.nn.functional (layout / padding guard).

The fixtures are SELF-GENERATED (oracle/make_golden.py): the reference ships none -> parity unpinned."""
import os

import numpy as np
import pytest
import torch

from oracle import facenet_oracle as fo
from tests.util_data import c1_images, structured_images, triplet_pool

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_variable_and_layer_counts_match_reference_shape_table():
    for E, tot, tr in ((128, 22808144, 22779312), (512, 23497424, 23467824)):
        p, trainable, reg = fo.build_params(E)
        assert fo.count_variables(p, trainable) == (tot, tr)
        assert len(reg) == 133                                   # 132 Conv2D + 1 Dense
        assert sum(k.endswith("/beta") for k in p) == 112        # 112 BatchNormalization layers
        assert sum(k.endswith("/bias") for k in p) == 21         # 21 biased `up` convolutions


def test_forward_macs_match_reference_shape_table():
    """1 401.17 M MAC per image at E=128: count them from the oracle's own tensor shapes."""
    p, _, _ = fo.build_params(128)
    o = fo.Oracle(p)
    macs = {"n": 0}
    orig = o._conv

    def counting(x, prefix, spec, bias=False):
        y = orig(x, prefix, spec, bias)
        kh, kw, cin, cout = p[prefix + "/kernel"].shape
        macs["n"] += y.shape[2] * y.shape[3] * kh * kw * cin * cout
        return y
    o._conv = counting
    o.forward(c1_images(1), training=False)
    macs["n"] += 1792 * 128
    assert abs(macs["n"] / 1e6 - 1401.17) < 0.02


def test_golden_c1_embeddings():
    g = np.load(os.path.join(GOLD, "c1_embeddings.npz"))
    x = c1_images()
    for E in (128, 512):
        for variant in ("fresh", "perturbed"):
            p, _, _ = fo.build_params(E, seed=0)
            if variant == "perturbed":
                fo.perturb_bn_stats(p, seed=1)
            e = fo.Oracle(p).forward(x, training=False).numpy()
            assert np.allclose(np.linalg.norm(e, axis=1), 1.0, atol=1e-5)
            assert np.abs(e - g[f"emb_{E}_{variant}"]).max() < 2e-5


def test_golden_image_processing_and_triplets():
    g = np.load(os.path.join(GOLD, "image_processing.npz"))
    xi = c1_images(3)
    xi[1] = 77
    for m in (0, 1):
        out = fo.image_processing(xi, m).numpy()
        assert np.allclose(out[:, ::16, ::16, :], g[f"mode{m}"], atol=1e-6)
    assert np.all(fo.image_processing(xi, 0).numpy()[1] == 0)           # constant image -> range clamps to eps, output 0
    with pytest.raises(ValueError):
        fo.image_processing(xi, 2)                                       # facenet.py:82
    t = np.load(os.path.join(GOLD, "triplets.npz"))
    emb, labels = triplet_pool()
    dist = fo.squared_distance_matrix(emb)
    assert np.allclose([dist.sum(dtype=np.float64), (dist ** 2).sum(dtype=np.float64)], t["dist_checksum"], rtol=1e-12)
    import hashlib
    assert np.array_equal(np.frombuffer(hashlib.sha1(np.ascontiguousarray(dist).tobytes()).digest(), dtype=np.uint8), t["dist_sha1"])
    e64 = emb.astype(np.float64)                                       # device-order fp32 matrix vs the definition in float64
    assert np.abs(dist - ((e64[:, None] - e64[None]) ** 2).sum(-1)).max() < 1e-6 and np.array_equal(dist, dist.T)
    for seed in (0, 7):
        for semi in (0, 1):
            sel = fo.select_triplets(dist, labels, 0.2, 30, seed, semi_hard=bool(semi))
            assert np.array_equal(sel, t[f"triplets_seed{seed}_semi{semi}"])
            a, p_, n = sel[:, 0], sel[:, 1], sel[:, 2]
            assert np.all(labels[a] == labels[p_]) and np.all(labels[a] != labels[n]) and np.all(a < p_)
            viol = dist[a, n] - dist[a, p_] < 0.2
            assert viol.sum() >= 20                                       # valid (margin-violating) triplets rank first ...
            assert np.all(viol[:viol.sum()]) and not np.any(viol[viol.sum():]) or semi   # ... then the top-up ones
            if semi:
                ok = viol & (dist[a, n] > dist[a, p_])
                assert ok.sum() >= 10 and np.all(ok[:ok.sum()])
    assert np.allclose(fo.pairwise_similarities(emb[:40], None, 0), t["pairwise_metric0_triu_first40"], atol=1e-6)


def test_pairwise_similarities_edge_cases():
    emb, _ = triplet_pool()
    s = fo.pairwise_similarities(emb[:5], emb[:7], 0)
    assert s.shape == (5, 7) and np.allclose(np.diag(s[:, :5]), 0, atol=1e-6)
    assert fo.pairwise_similarities(emb[:1], None).shape == (0,)        # statistics.py:32-36
    assert np.allclose(fo.pairwise_similarities(emb[:5], emb[:5], 1), np.arccos(np.clip(emb[:5] @ emb[:5].T, -1, 1)))
    with pytest.raises(ValueError):
        fo.pairwise_similarities(emb[:5] * 2, emb[:5] * 2)                # :40-42
    with pytest.raises(ValueError):
        fo.pairwise_similarities(emb[:5], emb[:5], metric=3)              # :55


def test_golden_training_trajectory_triplet():
    g = np.load(os.path.join(GOLD, "train_trajectory.npz"))
    params, trainable, regularized = fo.build_params(128, seed=0)
    xs = structured_images(9, seed=3)
    opt = fo.AdamKeras(trainable, params, lr=0.01)
    for step in range(2):
        data, total, grads, stats, _ = fo.train_step_grads(params, trainable, regularized, xs, "triplet", alpha=0.2)
        opt.step(params, grads)
        for k, v in stats.items():
            params[k].copy_(v)
        assert np.allclose([data, total], g["triplet_losses"][step], rtol=2e-3, atol=2e-4)
    assert total > data        # the Keras L2 term (5e-4 * sum w^2) is part of the total loss


# ---- independent NumPy direct loops (no torch.nn.functional) -----------------------------------------------
def _np_conv_nhwc(x, w_hwio, stride, same):
    n, h, wd, c = x.shape
    kh, kw, _, co = w_hwio.shape
    ph, pw = (kh // 2, kw // 2) if same else (0, 0)
    xp = np.zeros((n, h + 2 * ph, wd + 2 * pw, c), x.dtype)
    xp[:, ph:ph + h, pw:pw + wd] = x
    oh, ow = (h + 2 * ph - kh) // stride + 1, (wd + 2 * pw - kw) // stride + 1
    y = np.zeros((n, oh, ow, co), np.float64)
    for i in range(oh):
        for j in range(ow):
            patch = xp[:, i * stride:i * stride + kh, j * stride:j * stride + kw, :]
            y[:, i, j, :] = np.tensordot(patch, w_hwio, axes=([1, 2, 3], [0, 1, 2]))
    return y


@pytest.mark.parametrize("k,stride,same", [((3, 3), 2, False), ((3, 3), 1, True), ((1, 7), 1, True), ((7, 1), 1, True), ((1, 1), 1, True)])
def test_oracle_conv_bn_against_numpy_direct_loops(k, stride, same):
    rng = np.random.default_rng(1)
    x = rng.normal(size=(2, 9, 9, 5)).astype(np.float32)
    w = rng.normal(size=(k[0], k[1], 5, 4)).astype(np.float32)
    beta = rng.normal(size=4).astype(np.float32)
    p = {"L/kernel": torch.from_numpy(w), "L/bn/beta": torch.from_numpy(beta), "L/bn/moving_mean": torch.zeros(4), "L/bn/moving_variance": torch.ones(4)}
    o = fo.Oracle(p)
    spec = dict(name="L", cout=4, k=k, stride=stride, padding="same" if same else "valid")
    out = o._cbr(torch.from_numpy(x).permute(0, 3, 1, 2), "L", spec, training=True).permute(0, 2, 3, 1).numpy()
    y = _np_conv_nhwc(x.astype(np.float64), w.astype(np.float64), stride, same)
    mean, var = y.mean(axis=(0, 1, 2)), y.var(axis=(0, 1, 2))            # biased variance (hazard 3)
    ref = np.maximum((y - mean) / np.sqrt(var + 1e-3) + beta, 0)
    assert np.abs(out - ref).max() < 2e-5
    assert np.allclose(o.new_stats["L/bn/moving_mean"].numpy(), 0.01 * mean, atol=1e-6)
    assert np.allclose(o.new_stats["L/bn/moving_variance"].numpy(), 0.99 + 0.01 * var, atol=1e-6)


def test_adam_keras_form_and_lr_schedule():
    """Keras Adam: eps is added to the UN-corrected sqrt(v) (hazard 8) -- differs from torch.optim.Adam."""
    p = {"w": torch.tensor([1.0, -2.0])}
    opt = fo.AdamKeras(["w"], p, lr=0.05, epsilon=0.1)
    g = torch.tensor([0.5, -0.25])
    opt.step(p, {"w": g})
    m, v = 0.1 * g, 0.001 * g * g
    lr_t = 0.05 * np.sqrt(1 - 0.999) / (1 - 0.9)
    assert torch.allclose(p["w"], torch.tensor([1.0, -2.0]) - lr_t * m / (v.sqrt() + 0.1), atol=1e-7)
    sched = fo.LearningRateScheduler(schedule=[[100, 0.05], [200, 0.005], [300, 0.0005]])
    assert [sched(e) for e in (0, 99, 100, 199, 200, 299, 300, 1000)] == [0.05, 0.05, 0.005, 0.005, 0.0005, 0.0005, 0.0005, 0.0005]
    assert fo.LearningRateScheduler(value=0.1, schedule=[[1, 1.0]])(5) == 0.1


def test_hash_stream_is_fixed():
    """The counter-based RNG shared with csrc/loss.hip (lowbias32 fold): known answers."""
    assert fo.hash_u32(0, 0, 0) == fo.hash_u32(0, 0, 0)
    vals = [fo.hash_u32(1, 2, 3), fo.hash_u32(0, 0, 1), fo.hash_u32(123, 269, 2)]
    assert all(0 <= v < 2 ** 32 for v in vals) and len(set(vals)) == 3
    assert vals == [fo.hash_u32(1, 2, 3), fo.hash_u32(0, 0, 1), fo.hash_u32(123, 269, 2)]


def test_fma32_is_the_correctly_rounded_fused_multiply_add():
    """oracle._fma32 (what the device-order distance matrix is built from) against exact rational arithmetic, including a
    constructed double-rounding case: a*b + c whose float64 sum lands exactly on an fp32 midpoint."""
    from fractions import Fraction
    rng = np.random.default_rng(0)
    a, b, c = (rng.normal(size=3000).astype(np.float32) for _ in range(3))
    # a = 1 + 2^-12, b = 1 + 2^-12 : a*b = 1 + 2^-11 + 2^-24 (exact); c = 2^-60 pushes the sum just above the fp32 midpoint
    # 1 + 2^-11 + 2^-24 of (1 + 2^-11, 1 + 2^-11 + 2^-23); float64 drops c -> naive double rounding ties to even (down)
    a[0] = b[0] = np.float32(1 + 2.0 ** -12)
    c[0] = np.float32(2.0 ** -60)
    r = fo._fma32(a, b, c)
    for i in range(len(a)):
        exact = Fraction(float(a[i])) * Fraction(float(b[i])) + Fraction(float(c[i]))
        v = np.float32(r[i])
        for nb in (np.nextafter(v, np.float32(-np.inf)), np.nextafter(v, np.float32(np.inf))):
            assert abs(Fraction(float(v)) - exact) <= abs(Fraction(float(nb)) - exact), i
    assert r[0] == np.float32(1 + 2.0 ** -11 + 2.0 ** -23)            # rounded UP: the fused result, not the double-rounded one
    assert np.float32(np.float64(a[0]) * np.float64(b[0]) + np.float64(c[0])) != r[0]
