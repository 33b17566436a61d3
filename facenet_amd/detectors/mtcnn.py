"""MTCNN face detector on MI355X: drop-in for the PyPI `mtcnn` package the reference wraps (detectors/face_detector.py:63-78:
`from mtcnn.mtcnn import MTCNN; MTCNN().detect_faces(image)`), same constructor arguments, same result format.

Device work (libfacenet_hip.so, no CPU fallback): the image pyramid and the 24x24 / 48x48 candidate crops
(`fn_area_resize_frame` / `fn_area_resize_crop` = zero-padded crop + cv2 INTER_AREA + normalisation + transpose straight from
the uint8 frame, which stays resident in HBM), the P / R / O networks (`fn_conv2d_fwd` with bias + PReLU epilogue for every Conv2D and Dense,
`fn_maxpool2d_fwd`; both heads of a network are ONE convolution), and the P-Net softmax / threshold / compaction
(`fn_mtcnn_candidates`).  Stage 1 launches the whole pyramid back to back and synchronises once.
Greedy NMS: score order from the host's np.argsort, pairwise ratios + suppression scan on the device (`fn_nms_greedy`, float64).
Host work (NumPy, float64 like the package): box generation from the compacted cells, square-ing, regression.
Layout: activations [n][x][y][C] f16 (the networks see the transposed image), channels padded to multiples of 8 with zero
weights; weights packed [Cout][KH][KW][Cin]; a Dense after Flatten is a 'valid' convolution over the whole 3x3 map.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib

# name, KH, KW, Cin, Cout, prelu | pool: k, stride, same
_P = [("conv", "conv1", 3, 3, 3, 10, "prelu1"), ("pool", 2, 2, True), ("conv", "conv2", 3, 3, 10, 16, "prelu2"),
      ("conv", "conv3", 3, 3, 16, 32, "prelu3"), ("heads", ["conv4_1", "conv4_2"], 1, 1, 32, [2, 4])]
_R = [("conv", "conv1", 3, 3, 3, 28, "prelu1"), ("pool", 3, 2, True), ("conv", "conv2", 3, 3, 28, 48, "prelu2"), ("pool", 3, 2, False),
      ("conv", "conv3", 2, 2, 48, 64, "prelu3"), ("conv", "fc1", 3, 3, 64, 128, "prelu4"), ("heads", ["fc2_1", "fc2_2"], 1, 1, 128, [2, 4])]
_O = [("conv", "conv1", 3, 3, 3, 32, "prelu1"), ("pool", 3, 2, True), ("conv", "conv2", 3, 3, 32, 64, "prelu2"), ("pool", 3, 2, False),
      ("conv", "conv3", 3, 3, 64, 64, "prelu3"), ("pool", 2, 2, True), ("conv", "conv4", 2, 2, 64, 128, "prelu4"),
      ("conv", "fc1", 3, 3, 128, 256, "prelu5"), ("heads", ["fc2_1", "fc2_2", "fc2_3"], 1, 1, 256, [2, 4, 10])]
_SPECS = {"pnet": _P, "rnet": _R, "onet": _O}


def _pad8(c):
    return (c + 7) // 8 * 8


def _ptr(t):
    return t.data_ptr()


class InvalidImage(Exception):
    """Same name as the package's exception for an unusable input."""


class _Network:
    """One of the three networks as a fixed list of launches over cached activation buffers."""

    def __init__(self, net: str, weights: dict, device, dtype=torch.float16):
        self.net, self.device, self.dtype = net, device, dtype
        self.code = _lib.dtype_code(dtype)
        self.layers = []
        for item in _SPECS[net]:
            if item[0] == "pool":
                self.layers.append({"kind": "pool", "k": item[1], "stride": item[2], "same": item[3]})
                continue
            if item[0] == "conv":
                _, name, kh, kw, ci, co, pr = item
                kernel = np.asarray(weights[f"{net}/{name}/kernel"], np.float32).reshape(kh, kw, ci, co)   # Dense [kh*kw*ci, co] too
                bias = np.asarray(weights[f"{net}/{name}/bias"], np.float32)
                alpha = np.asarray(weights[f"{net}/{pr}/alpha"], np.float32).reshape(-1)
            else:
                _, names, kh, kw, ci, cos = item
                kernel = np.concatenate([np.asarray(weights[f"{net}/{n}/kernel"], np.float32).reshape(kh, kw, ci, c) for n, c in zip(names, cos)], axis=3)
                bias = np.concatenate([np.asarray(weights[f"{net}/{n}/bias"], np.float32) for n in names])
                co, alpha = sum(cos), None
            cip, cop = _pad8(ci), _pad8(co)
            pack = np.zeros((cop, kh, kw, cip), np.float32)
            pack[:co, :, :, :ci] = np.transpose(kernel, (3, 0, 1, 2))
            b = np.zeros(cop, np.float32)
            b[:co] = bias
            L = {"kind": "conv", "KH": kh, "KW": kw, "Cin": cip, "Cout": cop, "real_out": co,
                 "w": torch.from_numpy(pack).to(device=device, dtype=dtype).contiguous(),
                 "bias": torch.from_numpy(b).to(device), "f32": item[0] == "heads", "prelu": None}
            if alpha is not None:
                a = np.ones(cop, np.float32)
                a[:co] = alpha
                L["prelu"] = torch.from_numpy(a).to(device)
            self.layers.append(L)
        self.out_channels = self.layers[-1]["Cout"]
        self._plans = {}

    def _plan(self, key, cap, A, B):
        """Buffers + geometry for inputs [cap, A, B, 8]; `key` keeps pyramid levels of equal shape apart."""
        k = (key, cap, A, B)
        if k in self._plans:
            return self._plans[k]
        steps, a, b, c = [], A, B, 8
        x = torch.zeros(cap, A, B, 8, dtype=self.dtype, device=self.device)
        cur = x
        for L in self.layers:
            if L["kind"] == "pool":
                kk, s = L["k"], L["stride"]
                oa, ob = (-(-a // s), -(-b // s)) if L["same"] else ((a - kk) // s + 1, (b - kk) // s + 1)
                # Keras 'same': the padding (never a candidate for the maximum) is split with the smaller half in front
                pa = max((oa - 1) * s + kk - a, 0) // 2 if L["same"] else 0
                pb = max((ob - 1) * s + kk - b, 0) // 2 if L["same"] else 0
                y = torch.empty(cap, oa, ob, c, dtype=self.dtype, device=self.device)
                steps.append(("pool", L, cur, y, (a, b, c, oa, ob, pa, pb)))
            else:
                oa, ob = a - L["KH"] + 1, b - L["KW"] + 1
                if oa < 1 or ob < 1:
                    raise ValueError(f"mtcnn {self.net}: a {A}x{B} input is too small")
                y = torch.empty(cap, oa, ob, L["Cout"], dtype=torch.float32 if L["f32"] else self.dtype, device=self.device)
                steps.append(("conv", L, cur, y, (a, b, oa, ob)))
                c = L["Cout"]
            cur, a, b = y, oa, ob
        plan = {"x": x, "steps": steps, "out": cur}
        self._plans[k] = plan
        return plan

    def run(self, plan, n, stream):
        lib = _lib.load()
        for kind, L, src, dst, g in plan["steps"]:
            if kind == "pool":
                a, b, c, oa, ob, pa, pb = g
                _lib.check(lib.fn_maxpool2d_fwd(_ptr(src), c, _ptr(dst), c, n, a, b, c, L["k"], L["stride"], pa, pb, oa, ob, self.code, stream), "maxpool2d")
            else:
                a, b, oa, ob = g
                d = _lib.ConvDesc(N=n, H=a, W=b, Cin=L["Cin"], OH=oa, OW=ob, Cout=L["Cout"], KH=L["KH"], KW=L["KW"], stride=1, pad_h=0, pad_w=0,
                                  dtype=self.code, ld_x=L["Cin"], ld_y=L["Cout"], out_f32=1 if L["f32"] else 0, scale=1.0,
                                  x=_ptr(src), w=_ptr(L["w"]), y=_ptr(dst), bias=_ptr(L["bias"]),
                                  prelu=_ptr(L["prelu"]) if L["prelu"] is not None else None)
                _lib.check(lib.fn_conv2d_fwd(d, stream), f"mtcnn {self.net} conv")
        return plan["out"]


def _softmax2(logits):
    """Keras Softmax over the two class logits, fp32."""
    l = np.asarray(logits, np.float32)
    e = np.exp(l - l.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def _iou_keep(boxes, threshold, by_min):
    """Greedy NMS in score order (package's __nms): indices of the kept rows, best first."""
    if boxes.shape[0] == 0:
        return np.empty((0,), np.int64)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    area = (x2 - x1 + 1) * (y2 - y1 + 1)
    rest = np.argsort(boxes[:, 4])
    keep = []
    while rest.size:
        top, rest = rest[-1], rest[:-1]
        keep.append(top)
        iw = np.maximum(0.0, np.minimum(x2[top], x2[rest]) - np.maximum(x1[top], x1[rest]) + 1)
        ih = np.maximum(0.0, np.minimum(y2[top], y2[rest]) - np.maximum(y1[top], y1[rest]) + 1)
        inter = iw * ih
        with np.errstate(divide="ignore", invalid="ignore"):       # degenerate boxes: the ratio is inf / nan and fails `<=`
            ratio = inter / (np.minimum(area[top], area[rest]) if by_min else (area[top] + area[rest] - inter))
        rest = rest[ratio <= threshold]
    return np.asarray(keep, np.int64)


class _DeviceNms:
    """Greedy NMS jobs on the GPU (fn_nms_greedy): the score order is the host's np.argsort (what the package uses, ties and
    all), the pairwise ratios and the suppression scan run on the device in float64.  Several jobs share one upload, one
    workspace (stream order serialises them) and one download.  Jobs of <= HOST_MAX boxes, or beyond MAX_N (bit matrix of
    n^2 / 8 bytes), take the NumPy loop, which computes the same thing."""
    HOST_MAX, MAX_N = 16, 1 << 15

    def __init__(self, device):
        self.device = device
        self.ws = None

    def run(self, jobs):
        """jobs: [(boxes float64 [n, >=5], threshold, by_min)] -> [int64 indices of the kept boxes, best first]."""
        lib, st = _lib.load(), torch.cuda.current_stream(self.device).cuda_stream
        out = [None] * len(jobs)
        dev = [k for k, (b, _, _) in enumerate(jobs) if self.HOST_MAX < b.shape[0] <= self.MAX_N]
        for k, (b, thr, by_min) in enumerate(jobs):
            if k not in dev:
                out[k] = _iou_keep(b, thr, by_min)
        if not dev:
            return out
        sizes = np.asarray([jobs[k][0].shape[0] for k in dev], np.int32)
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        total = int(offs[-1])
        # one upload: float64 boxes (x1, y1, x2, y2, score) followed by the int32 score orders
        blob = np.empty(total * 40 + total * 4, np.uint8)
        blob[:total * 40].view(np.float64).reshape(total, 5)[:] = np.concatenate([jobs[k][0][:, :5] for k in dev], axis=0)
        blob[total * 40:].view(np.int32)[:] = np.concatenate([np.argsort(jobs[k][0][:, 4]) for k in dev])
        need = int(sum(int(n) * ((int(n) + 63) // 64) * 8 for n in sizes))
        if self.ws is None or self.ws.numel() < need:
            self.ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        d_blob = torch.from_numpy(blob).to(self.device)
        d_keep = torch.empty(total + len(dev), dtype=torch.int32, device=self.device)
        thr = np.asarray([jobs[k][1] for k in dev], np.float64)
        by_min = np.asarray([1 if jobs[k][2] else 0 for k in dev], np.int32)
        _lib.check(lib.fn_nms_greedy_batch(d_blob.data_ptr(), 5, d_blob.data_ptr() + total * 40, sizes.ctypes.data, thr.ctypes.data, by_min.ctypes.data, len(dev),
                                           self.ws.data_ptr(), self.ws.numel(), d_keep.data_ptr(), d_keep.data_ptr() + total * 4, st), "nms_greedy")
        keep = d_keep.cpu().numpy()
        for j, k in enumerate(dev):
            o = int(offs[j])
            out[k] = keep[o:o + int(keep[total + j])].astype(np.int64)
        return out


def _square(b):
    """Grow every box to a square around its centre (package's __rerec)."""
    w, h = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    side = np.maximum(w, h)
    b[:, 0] = b[:, 0] + w * 0.5 - side * 0.5
    b[:, 1] = b[:, 1] + h * 0.5 - side * 0.5
    b[:, 2] = b[:, 0] + side
    b[:, 3] = b[:, 1] + side
    return b


def _regress(b, reg):
    w, h = b[:, 2] - b[:, 0] + 1, b[:, 3] - b[:, 1] + 1
    out = b.copy()
    out[:, 0] = b[:, 0] + reg[:, 0] * w
    out[:, 1] = b[:, 1] + reg[:, 1] * h
    out[:, 2] = b[:, 2] + reg[:, 2] * w
    out[:, 3] = b[:, 3] + reg[:, 3] * h
    return out


def _windows(b):
    """1-based integer boxes -> (ox, oy, cw, ch) crop windows of the frame (package's __pad: zero outside the frame)."""
    x1, y1 = b[:, 0].astype(np.int32), b[:, 1].astype(np.int32)
    cw, ch = (b[:, 2] - b[:, 0] + 1).astype(np.int32), (b[:, 3] - b[:, 1] + 1).astype(np.int32)
    return np.ascontiguousarray(np.stack([x1 - 1, y1 - 1, cw, ch], axis=1), dtype=np.int32)


class MTCNN:
    """`MTCNN(weights_file=None, min_face_size=20, steps_threshold=None, scale_factor=0.709).detect_faces(img)`.

    weights_file: an `.npz` with the Keras variables `<net>/<layer>/{kernel,bias,alpha}` (oracle/mtcnn_oracle.py lists names and
    shapes), or the package's own `mtcnn_weights.npy` ({'pnet': [...], 'rnet': [...], 'onet': [...]} in get_weights() order);
    `weights` passes the dict directly.  There is no bundled trained file (no network here): one of the two is required.
    """

    def __init__(self, weights_file=None, min_face_size: int = 20, steps_threshold=None, scale_factor: float = 0.709, weights=None,
                 device="cuda:0", max_candidates: int = 1 << 13):
        if steps_threshold is None:
            steps_threshold = [0.6, 0.7, 0.7]
        if weights is None:
            if weights_file is None:
                raise ValueError("MTCNN: pass weights_file= or weights= (no trained weight file is bundled)")
            weights = load_weights(weights_file)
        _lib.load()
        self._min_face_size, self._steps_threshold, self._scale_factor = min_face_size, list(steps_threshold), scale_factor
        self.device = torch.device(device)
        self._nets = {n: _Network(n, weights, self.device) for n in ("pnet", "rnet", "onet")}
        self._max_cand = int(max_candidates)
        self._nms = _DeviceNms(self.device)
        self._pyr = {}

    @property
    def min_face_size(self):
        return self._min_face_size

    @min_face_size.setter
    def min_face_size(self, mfc=20):
        try:
            self._min_face_size = int(mfc)
        except ValueError:
            self._min_face_size = 20

    # -- stage 1 -------------------------------------------------------------------------------------------------------
    def _scales(self, height, width):
        m = 12 / self._min_face_size
        min_layer = np.amin([height, width]) * m
        scales, count = [], 0
        while min_layer >= 12:
            scales.append(m * np.power(self._scale_factor, count))
            min_layer = min_layer * self._scale_factor
            count += 1
        return scales

    def _pyramid(self, height, width):
        key = (height, width, self._min_face_size, self._scale_factor)
        if key not in self._pyr:
            levels = []
            pnet = self._nets["pnet"]
            for i, scale in enumerate(self._scales(height, width)):
                ws, hs = int(np.ceil(width * scale)), int(np.ceil(height * scale))
                if ws > width or hs > height:
                    raise _lib.FacenetHipError("mtcnn: min_face_size < 12 enlarges the uint8 frame; that cv2 path is not built")
                plan = pnet._plan(("pyr", i), 1, ws, hs)
                oa, ob = plan["out"].shape[1:3]
                levels.append({"scale": scale, "ws": ws, "hs": hs, "plan": plan, "A": oa, "B": ob})
            # one record buffer for all levels; row 0 is the header (word 0 = hit counter), so header + records leave in one copy
            self._pyr[key] = {"levels": levels, "rec": torch.zeros(1 + self._max_cand, 8, dtype=torch.float32, device=self.device),
                              "rows": torch.empty(height * max([lv["ws"] for lv in levels] + [1]) * 3, dtype=torch.float32, device=self.device)}
        return self._pyr[key]

    def pnet_maps(self, frame, level):
        """Testing aid: (offsets [A,B,4], face probability [A,B]) of one pyramid level in the network's [x][y] orientation."""
        st = torch.cuda.current_stream(self.device).cuda_stream
        H, W = frame.shape[:2]
        pyr = self._pyramid(H, W)
        lv = pyr["levels"][level]
        rec = torch.zeros(1 + lv["A"] * lv["B"], 8, dtype=torch.float32, device=self.device)
        self._run_level(frame, pyr, level, st, rec, threshold=-1.0, reset=True)     # threshold below every probability: all cells
        c = rec[1:].cpu().numpy()
        c = c[np.argsort(c[:, 0].copy().view(np.int32))]
        return c[:, 3:7].reshape(lv["A"], lv["B"], 4), c[:, 2].reshape(lv["A"], lv["B"])

    def _run_level(self, frame, pyr, i, st, rec, threshold=None, reset=False):
        lib = _lib.load()
        lv = pyr["levels"][i]
        H, W = frame.shape[:2]
        pnet = self._nets["pnet"]
        _lib.check(lib.fn_area_resize_frame(_ptr(frame), H, W, lv["hs"], lv["ws"], _ptr(pyr["rows"]), _ptr(lv["plan"]["x"]), pnet.code, st), "area_resize_frame")
        out = pnet.run(lv["plan"], 1, st)
        _lib.check(lib.fn_mtcnn_candidates(_ptr(out), lv["A"] * lv["B"], pnet.out_channels, float(self._steps_threshold[0] if threshold is None else threshold),
                                           _ptr(rec) + 32, _ptr(rec), rec.shape[0] - 1, i, 1 if reset else 0, st), "mtcnn_candidates")
        return out

    def _stage1(self, frame):
        H, W = frame.shape[:2]
        pyr = self._pyramid(H, W)
        levels = pyr["levels"]
        if not levels:
            return np.empty((0, 9))
        while True:
            # the pyramid is a fixed launch sequence per frame size (11 levels x 9 launches at 1280x720): captured once as a
            # HIP graph over a resident copy of the frame, replayed per frame
            if "graph" not in pyr:
                if "frame" not in pyr:
                    pyr["frame"] = torch.empty_like(frame)
                pyr["frame"].copy_(frame)
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    st = torch.cuda.current_stream(self.device).cuda_stream
                    pyr["outs"] = [self._run_level(pyr["frame"], pyr, i, st, pyr["rec"], reset=(i == 0)) for i in range(len(levels))]
                pyr["graph"] = g
            pyr["frame"].copy_(frame)
            pyr["graph"].replay()
            outs = pyr["outs"]
            first = min(pyr["rec"].shape[0], 1 + 2048)
            head = pyr["rec"][:first].cpu().numpy()          # the one synchronisation of the launches above: counter + first records
            n = int(head[0, 0:1].view(np.int32)[0])
            if n <= pyr["rec"].shape[0] - 1:
                break
            # more hits than room: the levels are deterministic, run them again into a buffer that fits
            pyr["rec"] = torch.zeros(1 + (1 << int(n - 1).bit_length()), 8, dtype=torch.float32, device=self.device)
            del pyr["graph"]
        c = head[1:1 + n] if n < first else np.concatenate([head[1:], pyr["rec"][first:1 + n].cpu().numpy()], axis=0)
        cell, tag = c[:, 0].copy().view(np.int32), c[:, 1].copy().view(np.int32)
        order = np.lexsort((cell, tag))                       # compaction order is arbitrary: back to level, then the package's scan order
        c, cell, tag = c[order], cell[order], tag[order]
        bounds = np.searchsorted(tag, np.arange(len(levels) + 1))
        per_level = []
        for i, lv in enumerate(levels):
            lo, hi = int(bounds[i]), int(bounds[i + 1])
            if hi == lo:
                continue
            a, b = cell[lo:hi] // lv["B"], cell[lo:hi] % lv["B"]            # a: image x cell, b: image y cell
            reg = c[lo:hi, 3:7]
            if hi - lo == 1:   # package quirk: with exactly one cell the offsets are read from the map flipped along x
                reg = outs[i][0, lv["A"] - 1 - int(a[0]), int(b[0]), 2:6].cpu().numpy()[None]
            bb = np.stack([a, b], axis=1)
            q1 = np.fix((2 * bb + 1) / lv["scale"])
            q2 = np.fix((2 * bb + 12) / lv["scale"])
            per_level.append(np.hstack([q1, q2, c[lo:hi, 2:3], reg]))
        keeps = self._nms.run([(boxes, 0.5, False) for boxes in per_level])
        total = np.concatenate([np.empty((0, 9))] + [boxes[keep] for boxes, keep in zip(per_level, keeps)], axis=0)
        if total.shape[0] == 0:
            return total
        total = total[self._nms.run([(total, 0.7, False)])[0]]
        rw, rh = total[:, 2] - total[:, 0], total[:, 3] - total[:, 1]
        total = np.stack([total[:, 0] + total[:, 5] * rw, total[:, 1] + total[:, 6] * rh, total[:, 2] + total[:, 7] * rw,
                          total[:, 3] + total[:, 8] * rh, total[:, 4]], axis=1)
        total = _square(total)
        total[:, 0:4] = np.fix(total[:, 0:4]).astype(np.int32)
        return total

    # -- stages 2 and 3 ------------------------------------------------------------------------------------------------
    def _refine(self, frame, boxes, net, size):
        """Crops of `boxes` -> network -> fp32 head rows [n, C] on the host."""
        lib, st = _lib.load(), torch.cuda.current_stream(self.device).cuda_stream
        H, W = frame.shape[:2]
        n = boxes.shape[0]
        cap = 64
        while cap < n:
            cap *= 2
        network = self._nets[net]
        plan = network._plan("crops", cap, size, size)
        win = torch.from_numpy(_windows(boxes)).to(self.device)
        _lib.check(lib.fn_area_resize_crop(_ptr(frame), H, W, _ptr(win), n, size, size, 0, _ptr(plan["x"]), network.code, st), "area_resize_crop")
        out = network.run(plan, n, st)
        return out[:n].reshape(n, -1).cpu().numpy()

    def _stage2(self, frame, total):
        if total.shape[0] == 0:
            return total
        rows = self._refine(frame, total, "rnet", 24)
        score = _softmax2(rows[:, 0:2])[:, 1]
        ok = np.where(score > self._steps_threshold[1])[0]
        total = np.hstack([total[ok, 0:4], score[ok, None]])
        reg = rows[ok, 2:6]
        if total.shape[0] > 0:
            keep = self._nms.run([(total, 0.7, False)])[0]
            total = _square(_regress(total[keep], reg[keep]))
        return total

    def _stage3(self, frame, total):
        if total.shape[0] == 0:
            return np.empty((0, 5)), np.empty((10, 0), np.float32)
        total = np.fix(total).astype(np.int32)
        rows = self._refine(frame, total, "onet", 48)
        score = _softmax2(rows[:, 0:2])[:, 1]
        ok = np.where(score > self._steps_threshold[2])[0]
        points = np.array(rows[ok, 6:16].T)                   # float32 [10, n] like the package's `points`
        total = np.hstack([total[ok, 0:4], score[ok, None]])
        reg = rows[ok, 2:6]
        w, h = total[:, 2] - total[:, 0] + 1, total[:, 3] - total[:, 1] + 1
        points[0:5, :] = np.tile(w, (5, 1)) * points[0:5, :] + np.tile(total[:, 0], (5, 1)) - 1
        points[5:10, :] = np.tile(h, (5, 1)) * points[5:10, :] + np.tile(total[:, 1], (5, 1)) - 1
        if total.shape[0] > 0:
            total = _regress(total, reg)
            keep = self._nms.run([(total, 0.7, True)])[0]
            total, points = total[keep], points[:, keep]
        return total, points

    def detect_boxes(self, img, trace=None):
        """-> (float64 [n, 5] = x1, y1, x2, y2, confidence; float32 [10, n] landmarks): the arrays behind detect_faces."""
        if img is None or not hasattr(img, "shape"):
            raise InvalidImage("Image not valid.")
        if isinstance(img, torch.Tensor):
            frame = img.to(device=self.device, dtype=torch.uint8).contiguous()
        else:
            arr = np.asarray(img)
            if arr.dtype != np.uint8:
                raise InvalidImage("Image not valid: uint8 pixels expected.")
            arr = np.ascontiguousarray(arr)
            frame = torch.from_numpy(arr if arr.flags.writeable else arr.copy()).to(self.device)
        if frame.dim() != 3 or frame.shape[2] != 3:
            raise InvalidImage("Image not valid: [height, width, 3] expected.")
        s1 = self._stage1(frame)
        s2 = self._stage2(frame, s1)
        s3, points = self._stage3(frame, s2)
        if trace is not None:
            trace.update(stage1=s1, stage2=s2, stage3=s3, points=points)
        return s3, points

    def detect_faces(self, img) -> list:
        total, points = self.detect_boxes(img)
        faces = []
        for box, kp in zip(total, points.T):
            x, y = max(0, int(box[0])), max(0, int(box[1]))
            faces.append({"box": [x, y, int(box[2] - x), int(box[3] - y)], "confidence": box[-1],
                          "keypoints": {"left_eye": (int(kp[0]), int(kp[5])), "right_eye": (int(kp[1]), int(kp[6])),
                                        "nose": (int(kp[2]), int(kp[7])), "mouth_left": (int(kp[3]), int(kp[8])),
                                        "mouth_right": (int(kp[4]), int(kp[9]))}})
        return faces


_ORDER = {
    "pnet": ["conv1", "prelu1", "conv2", "prelu2", "conv3", "prelu3", "conv4_1", "conv4_2"],
    "rnet": ["conv1", "prelu1", "conv2", "prelu2", "conv3", "prelu3", "fc1", "prelu4", "fc2_1", "fc2_2"],
    "onet": ["conv1", "prelu1", "conv2", "prelu2", "conv3", "prelu3", "conv4", "prelu4", "fc1", "prelu5", "fc2_1", "fc2_2", "fc2_3"],
}


def weights_from_lists(lists: dict) -> dict:
    """{'pnet': [arrays in Keras get_weights() order], ...} (the package's weight file) -> named variables."""
    out = {}
    for net, names in _ORDER.items():
        it = iter(lists[net])
        for name in names:
            if name.startswith("prelu"):
                out[f"{net}/{name}/alpha"] = np.asarray(next(it), np.float32)
            else:
                out[f"{net}/{name}/kernel"] = np.asarray(next(it), np.float32)
                out[f"{net}/{name}/bias"] = np.asarray(next(it), np.float32)
    return out


def load_weights(path) -> dict:
    """Named variables from an ``.npz`` (numeric arrays only, ``allow_pickle=False``: nothing in the file is executed).
    The PyPI package's own ``mtcnn_weights.npy`` is a PICKLED dict of lists; loading a pickle runs code from the file, so
    it is not accepted here -- convert it once, offline and explicitly, with ``tools/convert_mtcnn_weights.py`` (restricted
    unpickler: ndarray / list / dict only) and point ``mtcnn.weights_file`` at the resulting ``.npz``."""
    path = str(path)
    if not path.endswith(".npz"):
        raise ValueError(f"MTCNN weights must be an .npz of named variables, got {path!r}; convert the package's pickled "
                         f"mtcnn_weights.npy with tools/convert_mtcnn_weights.py")
    with np.load(path, allow_pickle=False) as z:
        return {k: np.asarray(z[k]) for k in z.files}
