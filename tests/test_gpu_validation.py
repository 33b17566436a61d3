"""GPU face-to-face validation (facenet_amd.statistics.ConfidenceMatrix / FaceToFaceValidation) against the NumPy
restatement of facenet/statistics.py:82-313 (oracle/statistics_oracle.py; parity unpinned by the reference)."""
import numpy as np
import pytest
import torch

from facenet_amd.config import Config
from facenet_amd.statistics import ConfidenceMatrix, FaceToFaceValidation, SimilarityCalculator
from oracle import statistics_oracle as so

pytestmark = pytest.mark.gpu


def _pool(sizes, E, seed, spread=0.5):
    rng = np.random.default_rng(seed)
    emb, labels = [], []
    for c, n in enumerate(sizes):
        cen = rng.normal(size=(1, E))
        emb.append(cen * 0.6 + rng.normal(size=(n, E)) * spread)
        labels += [c * 3 + 7] * n                      # non-contiguous label values
    emb = np.concatenate(emb).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    labels = np.array(labels)
    p = rng.permutation(len(labels))                   # unsorted on purpose
    return emb[p], labels[p]


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("sizes,E", [([5, 1, 9, 33, 2, 40, 7], 128), ([3] * 20, 512), ([70, 45], 96)])
def test_confidence_matrix_matches_reference_loops(sizes, E, metric):
    emb, labels = _pool(sizes, E, seed=len(sizes))
    upper = 4 if metric == 0 else np.pi
    thr = np.linspace(0, upper, 100)
    ref = so.ConfidenceMatrix(so.SimilarityCalculator(emb, labels, metric), thr)
    got = ConfidenceMatrix(SimilarityCalculator(emb, labels, metric), thr)
    # fp32 dot products differ in the last bits from NumPy's BLAS: a distance sitting on a threshold may fall on the
    # other side -> allow a few pair-counts of slack, expressed in the class-balanced units
    slack = 3.0 / max(1, min(n * (n - 1) // 2 for n in sizes if n > 1)) / len(sizes)
    for name in ("tp", "tn", "fp", "fn"):
        assert np.allclose(getattr(got, name), getattr(ref, name), atol=max(slack, 1e-9)), name
    for name in ("accuracy", "precision", "tp_rates", "tn_rates", "fp_rates"):
        assert np.allclose(getattr(got, name), getattr(ref, name), atol=5e-3), name
    # a single scalar threshold (the test-fold call, statistics.py:307-308)
    one = ConfidenceMatrix(SimilarityCalculator(emb, labels, metric), 1.1)
    ref1 = so.ConfidenceMatrix(so.SimilarityCalculator(emb, labels, metric), 1.1)
    assert np.allclose(one.accuracy, ref1.accuracy, atol=5e-3) and one.threshold.shape == (1,)


def test_errors_follow_the_reference():
    emb, labels = _pool([4, 4, 4], 64, seed=1)
    with pytest.raises(ValueError):
        ConfidenceMatrix(SimilarityCalculator(emb * 1.3, labels, 0), [0.5, 1.0])          # statistics.py:40-42
    with pytest.raises(ValueError):
        FaceToFaceValidation(emb, labels, Config({"metric": 2, "nrof_folds": 2, "far_target": 1e-3}))   # :258-260
    sims, w = SimilarityCalculator(emb, labels, 0).evaluate(1, 1)
    assert sims.shape == (6,) and w == 6 * 3


def test_face_to_face_validation_matches_reference():
    emb, labels = _pool([6] * 12 + [9, 3, 14], 64, seed=5)
    cfg = Config({"metric": 0, "nrof_folds": 4, "far_target": 1e-3})
    got = FaceToFaceValidation(emb, labels, cfg)
    ref = so.face_to_face_validation(emb, labels, 0, nrof_folds=4, far_target=1e-3)
    for crit, d in ref.items():
        for key, val in d.items():
            assert abs(got.dict[crit][key] - val) < 6e-3, (crit, key, got.dict[crit][key], val)
    text = repr(got)
    assert "MaximumAccuracy" in text and "Area under curve (AUC)" in text and "elapsed_time" in text
