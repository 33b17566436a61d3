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
def mean(x):
    return np.mean(np.array(x))


def std(x):
    return np.std(np.array(x))


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


class ConfidenceMatrix:
    """statistics.py:111-175: tp / tn / fp / fn per threshold with class-balanced weights, and the derived rates."""

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
        o = out.cpu().numpy().reshape(4, thr.size)
        lo, hi = (_decode_ord(v) for v in rng.cpu().tolist())
        if hi >= lo:                                   # at least one pair was evaluated
            lim = 1 + atol
            if lo < -lim or hi > lim:                  # statistics.py:40-42
                raise ValueError("\nembeddings must be normalized to 1, range {} {}".format(lo, hi))
        self.tp, self.tn, self.fp, self.fn = o[0].copy(), o[1].copy(), o[2].copy(), o[3].copy()

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
        tp_rates = np.ones(self.threshold.size)
        tp_rates[i] = self.tp[i] / (self.tp[i] + self.fn[i])
        return tp_rates

    @property
    def tn_rates(self):
        i = (self.tn + self.fp) > 0
        tn_rates = np.ones(self.threshold.size)
        tn_rates[i] = self.tn[i] / (self.tn[i] + self.fp[i])
        return tn_rates

    @property
    def fp_rates(self):
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


class Report:
    """statistics.py:178-234."""

    def __init__(self, criterion=None):
        self.criterion = criterion
        self.conf_matrix_train = []
        self.conf_matrix_test = []

    def __repr__(self):
        dct = self.dict
        info = self.criterion + '\n'
        info += ('Area under curve (AUC): {:1.5f}\n'.format(dct['auc']) +
                 'Equal error rate (EER): {:1.5f}\n'.format(dct['eer']) + '\n')
        info += ('Accuracy:  {:2.5f}+-{:2.5f}\n'.format(dct['accuracy'], dct['accuracy_std']) +
                 'Precision: {:2.5f}+-{:2.5f}\n'.format(dct['precision'], std(dct['precision_std'])) +
                 'Sensitivity (TPR, 1-a type 1 error): {:2.5f}+-{:2.5f}\n'.format(dct['tp_rates'], dct['tp_rates_std']) +
                 'Specificity (TNR, 1-b type 2 error): {:2.5f}+-{:2.5f}\n'.format(dct['tn_rates'], dct['tn_rates_std']) +
                 'Threshold: {:2.5f}+-{:2.5f}\n'.format(dct['threshold'], dct['threshold_std']) + '\n')
        return info

    def append_fold(self, name, conf_matrix):
        if name == 'train':
            self.conf_matrix_train.append(conf_matrix)
        else:
            self.conf_matrix_test.append(conf_matrix)

    @property
    def dict(self):
        import sklearn.metrics
        from scipy import interpolate
        from scipy.optimize import brentq
        tp_rates = np.mean(np.array([m.tp_rates for m in self.conf_matrix_train]), axis=0)
        tn_rates = np.mean(np.array([m.tn_rates for m in self.conf_matrix_train]), axis=0)
        dct = {'auc': -1, 'eer': -1}
        try:
            dct['auc'] = sklearn.metrics.auc(1 - tn_rates, tp_rates)
        except Exception:
            pass
        try:
            dct['eer'] = brentq(lambda x: 1. - x - interpolate.interp1d(1 - tn_rates, tp_rates)(x), 0., 1.)
        except Exception:
            pass

        def get(name):
            return [m.__getattribute__(name) for m in self.conf_matrix_test]

        for key in ('accuracy', 'precision', 'tp_rates', 'tn_rates', 'threshold'):
            x = get(key)
            dct[key] = np.mean(x)
            dct[key + '_std'] = np.std(x)
        return dct


class FaceToFaceValidation:
    """statistics.py:237-331: k-fold (KFold(shuffle=True, random_state=0) over image indices) max-accuracy and FAR-target
    thresholds on the training folds, scored on the test folds.  ``config``: .metric, .nrof_folds, .far_target."""

    def __init__(self, embeddings, labels, config, device: str = "cuda"):
        import time
        self.elapsed_time = time.monotonic()
        self.labels = np.asarray(labels)
        emb = embeddings if torch.is_tensor(embeddings) else torch.as_tensor(np.asarray(embeddings))
        self.embeddings = emb.to(device=device, dtype=torch.float32)
        assert (self.embeddings.shape[0] == len(self.labels))
        self.config = config
        self.reports = None
        if self.config.metric == 0:
            upper_threshold = 4
        elif self.config.metric == 1:
            upper_threshold = np.pi
        else:
            raise ValueError('Undefined similarity metric {}'.format(self.config.metric))
        self.thresholds = np.linspace(0, upper_threshold, 100)
        self._evaluate()

    def __repr__(self):
        info = (f'{self.__class__.__name__}\n' + f'metric: {self.config.metric}\n\n')
        for r in self.reports:
            info += str(r)
        info += f'elapsed_time: {self.elapsed_time}\n'
        return info

    def _evaluate(self):
        import time
        from sklearn.model_selection import KFold
        k_fold = KFold(n_splits=self.config.nrof_folds, shuffle=True, random_state=0)
        indices = np.arange(len(self.labels))
        self.reports = (Report(criterion='MaximumAccuracy'),
                        Report(criterion='FalseAlarmRate(FAR = {})'.format(self.config.far_target)))
        dev = self.embeddings.device
        for fold_idx, (train_set, test_set) in enumerate(k_fold.split(indices)):
            calculator = SimilarityCalculator(self.embeddings[torch.as_tensor(train_set, device=dev)], self.labels[train_set],
                                              metric=self.config.metric, device=str(dev))
            matrix = ConfidenceMatrix(calculator, self.thresholds)
            for i in range(len(self.reports)):
                self.reports[i].append_fold('train', matrix)
            accuracy_threshold = self.thresholds[np.argmax(matrix.accuracy)]
            far_threshold = 0
            if np.max(matrix.fp_rates) >= self.config.far_target:
                far_threshold = far_threshold_slinear(matrix.fp_rates, self.thresholds, self.config.far_target)
            calculator = SimilarityCalculator(self.embeddings[torch.as_tensor(test_set, device=dev)], self.labels[test_set],
                                              metric=self.config.metric, device=str(dev))
            self.reports[0].append_fold('test', ConfidenceMatrix(calculator, accuracy_threshold))
            self.reports[1].append_fold('test', ConfidenceMatrix(calculator, far_threshold))
        self.elapsed_time = time.monotonic() - self.elapsed_time

    @property
    def dict(self):
        return {r.criterion: r.dict for r in self.reports}

    def write_report(self, file):
        import datetime
        from pathlib import Path
        file = Path(file).expanduser()
        with file.open('at') as f:
            f.write(64 * '-' + '\n')
            f.write('{} {}\n'.format(self.__class__.__name__, datetime.datetime.now()))
            f.write('metric: {}\n\n'.format(self.config.metric))
            for r in self.reports:
                f.write(str(r))
