"""CPU restatement of the reference's face-to-face validation statistics -- TEST INFRASTRUCTURE ONLY
(imported by tests/ and tools/bench_validation.py's cpu leg; never by facenet_amd).

Follows facenet/statistics.py line by line: split_embeddings :68-79, SimilarityCalculator :82-108 (class-balanced
weights :93-101), ConfidenceMatrix :111-175, Report :178-234, FaceToFaceValidation :237-313 (KFold(shuffle=True,
random_state=0) over image indices, 100 thresholds linspace(0, 4 | pi), max-accuracy threshold :296, FAR threshold by
'slinear' interpolation :299-302).  PARITY UNPINNED: the reference ships no fixture for it; the shipped reports
(models/*/logs/report.txt) need the private dataset and weights."""
from __future__ import annotations

import numpy as np

from oracle.facenet_oracle import pairwise_similarities


def split_embeddings(embeddings, labels):
    return [embeddings[label == labels] for label in np.unique(labels)]


class SimilarityCalculator:
    def __init__(self, embeddings, labels, metric=0):
        self.metric = metric
        self.embeddings = split_embeddings(embeddings, labels)

    def evaluate(self, i, k):
        nrof_positive_class_pairs = self.nrof_classes
        nrof_negative_class_pairs = self.nrof_classes * (self.nrof_classes - 1) / 2
        if i == k:
            sims = pairwise_similarities(self.embeddings[i], metric=self.metric)
            weight = sims.size * nrof_positive_class_pairs
        else:
            sims = pairwise_similarities(self.embeddings[i], self.embeddings[k], metric=self.metric)
            weight = sims.size * nrof_negative_class_pairs
        return sims, weight

    @property
    def nrof_classes(self):
        return len(self.embeddings)


class ConfidenceMatrix:
    def __init__(self, calculator, threshold):
        self.threshold = np.array(threshold, ndmin=1)
        self.tp = np.zeros(self.threshold.size)
        self.tn = np.zeros(self.threshold.size)
        self.fp = np.zeros(self.threshold.size)
        self.fn = np.zeros(self.threshold.size)
        for i in range(calculator.nrof_classes):
            for k in range(i + 1):
                sims, weight = calculator.evaluate(i, k)
                if sims.size < 1:
                    continue
                for n, threshold in enumerate(self.threshold):
                    count = np.count_nonzero(sims < threshold)
                    if i == k:
                        self.tp[n] += count / weight
                        self.fn[n] += (sims.size - count) / weight
                    else:
                        self.fp[n] += count / weight
                        self.tn[n] += (sims.size - count) / weight

    @property
    def accuracy(self):
        return (self.tp + self.tn) / (self.tp + self.fp + self.tn + self.fn)

    @property
    def precision(self):
        i = (self.tp + self.fp) > 0
        precision = np.ones(self.threshold.size)
        precision[i] = self.tp[i] / (self.tp[i] + self.fp[i])
        return precision

    @property
    def tp_rates(self):
        i = (self.tp + self.fn) > 0
        r = np.ones(self.threshold.size)
        r[i] = self.tp[i] / (self.tp[i] + self.fn[i])
        return r

    @property
    def tn_rates(self):
        i = (self.tn + self.fp) > 0
        r = np.ones(self.threshold.size)
        r[i] = self.tn[i] / (self.tn[i] + self.fp[i])
        return r

    @property
    def fp_rates(self):
        return 1 - self.tn_rates

    @property
    def fn_rates(self):
        return 1 - self.tp_rates


def far_threshold_slinear(fp_rates, thresholds, far_target):
    """statistics.py:299-302: ``interp1d(fp_rates, thresholds, kind='slinear')(far_target)``.  fp_rates is a monotone
    step function of the threshold with many repeated values; the scipy the reference ran (<= 1.5) accepted that, current
    scipy raises "Expect x to not have duplicates".  Piecewise-linear reading used here (and by facenet_amd): between
    the LAST threshold whose fp_rate <= far_target and the FIRST one above it."""
    fp = np.asarray(fp_rates, dtype=np.float64)
    thr = np.asarray(thresholds, dtype=np.float64)
    j = int(np.searchsorted(fp, far_target, side="right")) - 1
    if j < 0:
        return thr[0]
    if j >= len(fp) - 1:
        return thr[-1]
    if fp[j + 1] == fp[j]:
        return thr[j]
    return thr[j] + (far_target - fp[j]) / (fp[j + 1] - fp[j]) * (thr[j + 1] - thr[j])


def face_to_face_validation(embeddings, labels, metric=0, nrof_folds=10, far_target=1e-3):
    """FaceToFaceValidation._evaluate (:277-313) -> {criterion: dict} like FaceToFaceValidation.dict (:315-318)."""
    import sklearn.metrics
    from scipy import interpolate
    from scipy.optimize import brentq
    from sklearn.model_selection import KFold
    if metric == 0:
        upper = 4
    elif metric == 1:
        upper = np.pi
    else:
        raise ValueError("Undefined similarity metric {}".format(metric))
    thresholds = np.linspace(0, upper, 100)
    k_fold = KFold(n_splits=nrof_folds, shuffle=True, random_state=0)
    train_m, test_acc, test_far = [], [], []
    for train_set, test_set in k_fold.split(np.arange(len(labels))):
        m = ConfidenceMatrix(SimilarityCalculator(embeddings[train_set], labels[train_set], metric), thresholds)
        train_m.append(m)
        acc_thr = thresholds[np.argmax(m.accuracy)]
        far_thr = 0
        if np.max(m.fp_rates) >= far_target:
            far_thr = far_threshold_slinear(m.fp_rates, thresholds, far_target)
        calc = SimilarityCalculator(embeddings[test_set], labels[test_set], metric)
        test_acc.append(ConfidenceMatrix(calc, acc_thr))
        test_far.append(ConfidenceMatrix(calc, far_thr))

    def report(test):
        tp_rates = np.mean(np.array([m.tp_rates for m in train_m]), axis=0)
        tn_rates = np.mean(np.array([m.tn_rates for m in train_m]), axis=0)
        dct = {"auc": -1, "eer": -1}
        try:
            dct["auc"] = sklearn.metrics.auc(1 - tn_rates, tp_rates)
        except Exception:
            pass
        try:
            dct["eer"] = brentq(lambda x: 1. - x - interpolate.interp1d(1 - tn_rates, tp_rates)(x), 0., 1.)
        except Exception:
            pass
        for key in ("accuracy", "precision", "tp_rates", "tn_rates", "threshold"):
            x = [getattr(m, key) for m in test]
            dct[key] = np.mean(x)
            dct[key + "_std"] = np.std(x)
        return dct
    return {"MaximumAccuracy": report(test_acc), "FalseAlarmRate(FAR = {})".format(far_target): report(test_far)}
