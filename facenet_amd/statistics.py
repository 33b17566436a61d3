"""``pairwise_similarities`` with the reference's signature and error behaviour (facenet/statistics.py:22-57),
computed by the wavefront-reduced fn_pairwise_sqdist kernel."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .engine import _ptr


def _decode_ord(i: int) -> float:
    i = int(i)
    bits = i if i >= 0 else (i ^ 0x7FFFFFFF)
    return float(np.array([bits & 0xFFFFFFFF], dtype=np.uint32).view(np.float32)[0])


def pairwise_similarities(xa, xb=None, metric: int = 0, atol: float = 1.e-5, device: str = "cuda"):
    """xa [n,E], xb [m,E] unit-norm rows -> 2(1 - xa.xb^T) (metric 0) or arccos (metric 1); with ``xb=None`` the strict
    upper triangle of xa against itself, flattened row-major (np.triu_indices order)."""
    lib = _lib.load()
    if metric not in (0, 1):
        raise ValueError("Undefined similarity metric {}".format(metric))      # statistics.py:55
    a = torch.as_tensor(np.asarray(xa) if not torch.is_tensor(xa) else xa).to(device=device, dtype=torch.float32).contiguous()
    b = a if xb is None else torch.as_tensor(np.asarray(xb) if not torch.is_tensor(xb) else xb).to(device=device, dtype=torch.float32).contiguous()
    n, m, E = a.shape[0], b.shape[0], a.shape[1]
    if n == 0 or m == 0:
        return np.zeros((0,) if xb is None else (n, m), dtype=np.float32)
    out = torch.empty(n, m, dtype=torch.float32, device=a.device)
    rng = torch.zeros(2, dtype=torch.int32, device=a.device)
    st = torch.cuda.current_stream(a.device).cuda_stream
    _lib.check(lib.fn_pairwise_sqdist(_ptr(a), _ptr(b), _ptr(out), _ptr(rng), n, m, E, metric, st), "pairwise_sqdist")
    if xb is None:
        iu = torch.triu_indices(n, n, offset=1, device=a.device)
        sims = out[iu[0], iu[1]]
        if sims.numel() == 0:
            return sims.cpu().numpy()
    else:
        sims = out
    lo, hi = (_decode_ord(v) for v in rng.cpu().tolist())
    lim = 1 + atol
    if lo < -lim or hi > lim:   # statistics.py:40-42 (the kernel reports min/max over the full matrix)
        raise ValueError("\nembeddings must be normalized to 1, range {} {}".format(lo, hi))
    return sims.cpu().numpy()


# ------------------------------------------------------------------------------------------------------------------
# Face-to-face validation (facenet/statistics.py:82-331) on the GPU.  Same class names, properties and report text as
# the reference; the O(classes^2 x thresholds) NumPy loops run as ONE launch of fn_confidence_counts per matrix.
# ------------------------------------------------------------------------------------------------------------------
class SimilarityCalculator:
    """statistics.py:82-108.  Holds the embeddings grouped by class on the device (rows sorted by label)."""

    def __init__(self, embeddings, labels, metric=0, device: str = "cuda"):
        self.metric = metric
        labels = np.asarray(labels)
        emb = embeddings if torch.is_tensor(embeddings) else torch.as_tensor(np.asarray(embeddings))
        order = np.argsort(labels, kind="stable")
        uniq, counts = np.unique(labels, return_counts=True)
        self.class_start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        self.emb = emb.to(device=device, dtype=torch.float32)[torch.as_tensor(order, device=device)].contiguous()
        self._cls = torch.as_tensor(self.class_start, device=self.emb.device)

    @property
    def nrof_classes(self):
        return len(self.class_start) - 1

    def nrof_images(self, i):
        return int(self.class_start[i + 1] - self.class_start[i])

    def evaluate(self, i, k):
        """statistics.py:92-103 for one class pair (host convenience; ConfidenceMatrix does not loop over it)."""
        a = self.emb[self.class_start[i]:self.class_start[i + 1]]
        if i == k:
            sims = pairwise_similarities(a, None, self.metric, device=str(self.emb.device))
            weight = sims.size * self.nrof_classes
        else:
            b = self.emb[self.class_start[k]:self.class_start[k + 1]]
            sims = pairwise_similarities(a, b, self.metric, device=str(self.emb.device))
            weight = sims.size * (self.nrof_classes * (self.nrof_classes - 1) / 2)
        return sims, weight


def _ratio_or_one(num, den):
    """num / den element-wise, 1 where den == 0 (an empty positive or negative set scores perfectly, statistics.py:144-168)."""
    num, den = np.asarray(num, dtype=np.float64), np.asarray(den, dtype=np.float64)
    return np.divide(num, den, out=np.ones_like(den), where=den > 0)


class ConfidenceMatrix:
    """Interface of statistics.py:111-175 (attributes tp / tn / fp / fn / threshold, the six rate properties).  The
    class-balanced counts for ALL thresholds come from one fn_confidence_counts launch into ``counts`` [4, T]; every rate is
    a ratio of two rows of it."""

    _ROWS = {"tp": 0, "tn": 1, "fp": 2, "fn": 3}

    def __init__(self, calculator: SimilarityCalculator, threshold, atol: float = 1.e-5):
        lib = _lib.load()
        self.threshold = np.array(threshold, ndmin=1)
        thr = self.threshold.astype(np.float32)
        if thr.size > 1 and not np.all(np.diff(thr) >= 0):
            raise ValueError("thresholds must be ascending")
        dev = calculator.emb.device
        t_dev = torch.as_tensor(thr, device=dev)
        out = torch.zeros(4 * thr.size, dtype=torch.float64, device=dev)
        rng = torch.zeros(2, dtype=torch.int32, device=dev)
        n, E = calculator.emb.shape
        st = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.fn_confidence_counts(_ptr(calculator.emb), _ptr(calculator._cls), calculator.nrof_classes, E, _ptr(t_dev), thr.size,
                                            calculator.metric, _ptr(out), _ptr(rng), st), "confidence_counts")
        self.counts = out.cpu().numpy().reshape(4, thr.size)
        lo, hi = (_decode_ord(v) for v in rng.cpu().tolist())
        if hi >= lo and (lo < -(1 + atol) or hi > 1 + atol):   # some pair was evaluated and left [-1, 1]: statistics.py:40-42
            raise ValueError("\nembeddings must be normalized to 1, range {} {}".format(lo, hi))

    def __getattr__(self, name):        # tp, tn, fp, fn are views of the count table
        row = ConfidenceMatrix._ROWS.get(name)
        if row is None or "counts" not in self.__dict__:
            raise AttributeError(name)
        return self.counts[row]

    @property
    def accuracy(self):
        return (self.tp + self.tn) / self.counts.sum(axis=0)

    @property
    def precision(self):
        return _ratio_or_one(self.tp, self.tp + self.fp)

    @property
    def tp_rates(self):      # sensitivity / recall
        return _ratio_or_one(self.tp, self.tp + self.fn)

    @property
    def tn_rates(self):      # specificity
        return _ratio_or_one(self.tn, self.tn + self.fp)

    @property
    def fp_rates(self):      # false alarm rate
        return 1 - self.tn_rates

    @property
    def fn_rates(self):
        return 1 - self.tp_rates


def far_threshold_slinear(fp_rates, thresholds, far_target):
    """statistics.py:299-302 ``interp1d(fp_rates, thresholds, kind='slinear')(far_target)``; fp_rates repeats values, which
    the scipy the reference ran accepted and current scipy rejects: piecewise-linear between the LAST threshold whose
    fp_rate <= far_target and the FIRST one above it."""
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


_SCORED = ("accuracy", "precision", "tp_rates", "tn_rates", "threshold")     # keys of Report.dict besides auc / eer
_REPORT_LINES = (("Accuracy:  ", "accuracy"), ("Precision: ", "precision"), ("Sensitivity (TPR, 1-a type 1 error): ", "tp_rates"),
                 ("Specificity (TNR, 1-b type 2 error): ", "tn_rates"), ("Threshold: ", "threshold"))


def _roc_summary(fpr, tpr):
    """(AUC, EER) of a ROC polyline; -1 where the reference's try/except leaves its default (statistics.py:212-222)."""
    import sklearn.metrics
    from scipy import interpolate
    from scipy.optimize import brentq
    auc = eer = -1
    try:
        auc = sklearn.metrics.auc(fpr, tpr)
    except Exception:
        pass
    try:
        roc = interpolate.interp1d(fpr, tpr)
        eer = brentq(lambda x: 1. - x - roc(x), 0., 1.)
    except Exception:
        pass
    return auc, eer


class Report:
    """Interface of statistics.py:178-234 (criterion, conf_matrix_train / conf_matrix_test, append_fold, dict, the report
    text).  The dictionary is computed from stacked [folds, ...] arrays; the text is rendered from a line table."""

    def __init__(self, criterion=None):
        self.criterion = criterion
        self.conf_matrix_train = []
        self.conf_matrix_test = []

    def append_fold(self, name, conf_matrix):
        (self.conf_matrix_train if name == 'train' else self.conf_matrix_test).append(conf_matrix)

    @property
    def dict(self):
        train = self.conf_matrix_train
        tpr = np.stack([m.tp_rates for m in train]).mean(axis=0)
        fpr = 1 - np.stack([m.tn_rates for m in train]).mean(axis=0)
        auc, eer = _roc_summary(fpr, tpr)
        out = {'auc': auc, 'eer': eer}
        for key in _SCORED:
            per_fold = np.array([getattr(m, key) for m in self.conf_matrix_test])
            out[key] = per_fold.mean()
            out[key + '_std'] = per_fold.std()
        return out

    def __repr__(self):
        d = self.dict
        head = '{}\nArea under curve (AUC): {:1.5f}\nEqual error rate (EER): {:1.5f}\n\n'.format(self.criterion, d['auc'], d['eer'])
        # the reference prints std(precision_std) -- the spread of a scalar, i.e. 0 -- on the precision line (statistics.py:196)
        spread = {k: (0.0 if k == 'precision' else d[k + '_std']) for _, k in _REPORT_LINES}
        body = ''.join('{}{:2.5f}+-{:2.5f}\n'.format(label, d[k], spread[k]) for label, k in _REPORT_LINES)
        return head + body + '\n'


class FaceToFaceValidation:
    """Interface of statistics.py:237-331: k-fold (KFold(shuffle=True, random_state=0) over image indices); per fold the
    max-accuracy and the FAR-target thresholds are chosen on the training part and scored on the held-out part.
    ``config``: .metric, .nrof_folds, .far_target.  Embeddings stay on the device; every matrix is one kernel launch."""

    _UPPER = {0: 4, 1: np.pi}       # largest possible similarity value per metric (statistics.py:253-258)

    def __init__(self, embeddings, labels, config, device: str = "cuda"):
        import time
        t0 = time.monotonic()
        self.config = config
        if config.metric not in self._UPPER:
            raise ValueError('Undefined similarity metric {}'.format(config.metric))
        self.labels = np.asarray(labels)
        emb = embeddings if torch.is_tensor(embeddings) else torch.as_tensor(np.asarray(embeddings))
        self.embeddings = emb.to(device=device, dtype=torch.float32)
        assert self.embeddings.shape[0] == len(self.labels)
        self.thresholds = np.linspace(0, self._UPPER[config.metric], 100)
        self.reports = (Report(criterion='MaximumAccuracy'),
                        Report(criterion='FalseAlarmRate(FAR = {})'.format(config.far_target)))
        self._evaluate()
        self.elapsed_time = time.monotonic() - t0

    def _calculator(self, subset) -> SimilarityCalculator:
        dev = self.embeddings.device
        return SimilarityCalculator(self.embeddings[torch.as_tensor(subset, device=dev)], self.labels[subset],
                                    metric=self.config.metric, device=str(dev))

    def _fold_thresholds(self, matrix: ConfidenceMatrix):
        """(threshold of maximal accuracy, threshold where the false-alarm rate reaches far_target or 0)."""
        best = self.thresholds[int(np.argmax(matrix.accuracy))]
        fpr = matrix.fp_rates
        far = far_threshold_slinear(fpr, self.thresholds, self.config.far_target) if fpr.max() >= self.config.far_target else 0
        return best, far

    def _evaluate(self):
        from sklearn.model_selection import KFold
        folds = KFold(n_splits=self.config.nrof_folds, shuffle=True, random_state=0)
        for train_set, test_set in folds.split(np.arange(len(self.labels))):
            fitted = ConfidenceMatrix(self._calculator(train_set), self.thresholds)
            held_out = self._calculator(test_set)
            for report, thr in zip(self.reports, self._fold_thresholds(fitted)):
                report.append_fold('train', fitted)
                report.append_fold('test', ConfidenceMatrix(held_out, thr))

    def _text(self, header: str, footer: str = '') -> str:
        return header + 'metric: {}\n\n'.format(self.config.metric) + ''.join(str(r) for r in self.reports) + footer

    def __repr__(self):
        return self._text(f'{self.__class__.__name__}\n', f'elapsed_time: {self.elapsed_time}\n')

    @property
    def dict(self):
        return {r.criterion: r.dict for r in self.reports}

    def write_report(self, file):
        import datetime
        from pathlib import Path
        with Path(file).expanduser().open('at') as f:
            f.write(self._text(64 * '-' + '\n' + '{} {}\n'.format(self.__class__.__name__, datetime.datetime.now())))
