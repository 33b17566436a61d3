"""CPU restatement of the MTCNN face-detection cascade (BASELINE.json config 5).

TEST INFRASTRUCTURE ONLY: imported by tests/ and tools/bench_mtcnn.py's CPU leg (nothing in facenet_amd/ imports it).
PARITY UNPINNED: the reference holds no detector arithmetic at all -- detectors/face_detector.py:63-78 is a 16-line wrapper
around the PyPI package `mtcnn` (ipazc/mtcnn, NOT pinned: it is not even listed in requirements.txt:1-14; the 0.1.x line of
2019-2020 is what `from mtcnn.mtcnn import MTCNN; MTCNN().detect_faces(image)` resolved to when the reference was written),
which in turn resizes with OpenCV (`cv2.resize(..., interpolation=cv2.INTER_AREA)`).  Neither package is installed here and
the reference ships no detector fixtures, so nothing pins this file; it restates the published algorithm:

  * Zhang et al., "Joint Face Detection and Alignment using Multi-task Cascaded Convolutional Networks" (2016): P-Net over an
    image pyramid (factor 0.709, min face 20 px, 12x12 cells at stride 2) -> NMS -> R-Net on 24x24 crops -> NMS -> O-Net on
    48x48 crops -> NMS; thresholds 0.6 / 0.7 / 0.7;
  * the package's Keras graphs (network/factory.py): Conv2D 'valid' + PReLU(shared_axes=[1, 2]) + MaxPooling2D as listed in
    NETS below, heads = Softmax over 2 classes, 4 box offsets, (O-Net) 10 landmark coordinates; the networks see the
    TRANSPOSED image ([x][y]) because the published weights come from a column-major framework;
  * its box arithmetic (detect_faces / __stage1..3, __generate_bounding_box, __nms, __rerec, __pad, __bbreg), including its
    quirks: 1-based crop coordinates, `np.fix`, the flipped regression lookup when exactly one cell fires, scores compared
    with >= in stage 1 and > in stages 2 / 3;
  * OpenCV's INTER_AREA (imgproc/resize.cpp): `computeResizeAreaTab` weights in float, float accumulators and round-half-even
    saturation for uint8 sources, double accumulators for float64 sources, and -- when either axis is enlarged -- the
    bilinear path with "area mode" coordinates.  Not restated: OpenCV's integer-ratio fast path (`ResizeAreaFast`, same
    average up to the rounding of 2x2 blocks) and its fixed-point bilinear for uint8 (stage 1 only ever shrinks).
Convolutions run in fp32 on the CPU through torch.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

# (kind, ...) per layer.  conv: (name, KH, KW, Cin, Cout); prelu: name; pool: (k, stride, 'same'|'valid'); dense: (name, in, out)
NETS = {
    "pnet": {"input": None,
             "body": [("conv", "conv1", 3, 3, 3, 10), ("prelu", "prelu1"), ("pool", 2, 2, "same"),
                      ("conv", "conv2", 3, 3, 10, 16), ("prelu", "prelu2"),
                      ("conv", "conv3", 3, 3, 16, 32), ("prelu", "prelu3")],
             "heads": [("conv", "conv4_1", 1, 1, 32, 2), ("conv", "conv4_2", 1, 1, 32, 4)]},
    "rnet": {"input": 24,
             "body": [("conv", "conv1", 3, 3, 3, 28), ("prelu", "prelu1"), ("pool", 3, 2, "same"),
                      ("conv", "conv2", 3, 3, 28, 48), ("prelu", "prelu2"), ("pool", 3, 2, "valid"),
                      ("conv", "conv3", 2, 2, 48, 64), ("prelu", "prelu3"),
                      ("dense", "fc1", 576, 128), ("prelu", "prelu4")],
             "heads": [("dense", "fc2_1", 128, 2), ("dense", "fc2_2", 128, 4)]},
    "onet": {"input": 48,
             "body": [("conv", "conv1", 3, 3, 3, 32), ("prelu", "prelu1"), ("pool", 3, 2, "same"),
                      ("conv", "conv2", 3, 3, 32, 64), ("prelu", "prelu2"), ("pool", 3, 2, "valid"),
                      ("conv", "conv3", 3, 3, 64, 64), ("prelu", "prelu3"), ("pool", 2, 2, "same"),
                      ("conv", "conv4", 2, 2, 64, 128), ("prelu", "prelu4"),
                      ("dense", "fc1", 1152, 256), ("prelu", "prelu5")],
             "heads": [("dense", "fc2_1", 256, 2), ("dense", "fc2_2", 256, 4), ("dense", "fc2_3", 256, 10)]},
}


def variable_shapes(net: str):
    """Keras variables of one network in `model.get_weights()` order: [(key, shape)], Keras layouts (HWIO kernels, [in,out] dense)."""
    out, prev_c, prev_kind = [], None, None
    spec = NETS[net]
    for item in spec["body"] + spec["heads"]:
        if item[0] == "conv":
            _, name, kh, kw, ci, co = item
            out += [(f"{net}/{name}/kernel", (kh, kw, ci, co)), (f"{net}/{name}/bias", (co,))]
            prev_c, prev_kind = co, "conv"
        elif item[0] == "dense":
            _, name, ci, co = item
            out += [(f"{net}/{name}/kernel", (ci, co)), (f"{net}/{name}/bias", (co,))]
            prev_c, prev_kind = co, "dense"
        elif item[0] == "prelu":
            # PReLU(shared_axes=[1, 2]) after a convolution: alpha [1, 1, C]; PReLU() after a dense layer: alpha [C]
            out.append((f"{net}/{item[1]}/alpha", (prev_c,) if prev_kind == "dense" else (1, 1, prev_c)))
    return out


def random_weights(seed: int = 0, face_bias=(0.0, 0.0, 0.0)):
    """Glorot-uniform kernels, small random biases, PReLU slopes in [0.1, 0.4]; `face_bias[i]` is added to the face logit of
    network i (lets a test or a benchmark choose how many cells / crops pass a stage).  Synthetic: the package's trained
    weight file is not available here."""
    rng = np.random.default_rng(seed)
    w = {}
    for ni, net in enumerate(("pnet", "rnet", "onet")):
        for key, shape in variable_shapes(net):
            if key.endswith("/kernel"):
                fan_in = int(np.prod(shape[:-1]))
                fan_out = shape[-1] * (int(np.prod(shape[:-2])) if len(shape) == 4 else 1)
                lim = np.sqrt(6.0 / (fan_in + fan_out))
                w[key] = rng.uniform(-lim, lim, shape).astype(np.float32)
            elif key.endswith("/bias"):
                w[key] = rng.uniform(-0.1, 0.1, shape).astype(np.float32)
            else:
                w[key] = rng.uniform(0.1, 0.4, shape).astype(np.float32)
        head = {"pnet": "conv4_1", "rnet": "fc2_1", "onet": "fc2_1"}[net]
        w[f"{net}/{head}/bias"][1] += np.float32(face_bias[ni])
    return w


# ------------------------------------------------------------------------------------------------------------------------
# cv2.resize(..., interpolation=cv2.INTER_AREA)
# ------------------------------------------------------------------------------------------------------------------------
def _area_tab(ssize: int, dsize: int, scale: float):
    """computeResizeAreaTab: list of (dst index, src index, float32 weight) in table order."""
    tab = []
    for d in range(dsize):
        fsx1 = d * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = int(np.ceil(fsx1)), int(np.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((d, sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((d, sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((d, sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def _area_linear_coords(ssize: int, dsize: int, scale: float, inv_scale: float, horizontal: bool):
    idx, frac = np.zeros(dsize, np.int64), np.zeros(dsize, np.float32)
    for d in range(dsize):
        s = int(np.floor(d * scale))
        f = np.float32((d + 1) - (s + 1) * inv_scale)
        f = np.float32(0.0) if f <= 0 else np.float32(f - np.floor(f))
        if horizontal:
            if s < 0:
                s, f = 0, np.float32(0.0)
            if s >= ssize - 1:
                s, f = ssize - 1, np.float32(0.0)
        else:
            s = min(max(s, 0), ssize - 1)
        idx[d], frac[d] = s, f
    return idx, frac


def resize_area(src: np.ndarray, width: int, height: int) -> np.ndarray:
    """uint8 [h,w,c] -> uint8, float64 [h,w,c] -> float64."""
    h, w = src.shape[:2]
    is_u8 = src.dtype == np.uint8
    wt = np.float32 if is_u8 else np.float64
    inv_sx, inv_sy = width / w, height / h
    scale_x, scale_y = 1.0 / inv_sx, 1.0 / inv_sy
    s = src.astype(wt)
    if scale_x >= 1.0 and scale_y >= 1.0:
        xtab, ytab = _area_tab(w, width, scale_x), _area_tab(h, height, scale_y)
        # horizontal pass of every source row: buf[dx] += S[sx] * alpha, in table order
        buf = np.zeros((h, width) + src.shape[2:], wt)
        for d, sx, a in xtab:
            buf[:, d] = buf[:, d] + s[:, sx] * wt(a)
        out = np.zeros((height, width) + src.shape[2:], wt)
        seen = set()
        for d, sy, b in ytab:
            if d in seen:
                out[d] = out[d] + wt(b) * buf[sy]
            else:
                out[d] = wt(b) * buf[sy]
                seen.add(d)
    else:
        if is_u8:
            raise NotImplementedError("INTER_AREA enlargement of a uint8 image (fixed-point bilinear) is not restated")
        xi, fx = _area_linear_coords(w, width, scale_x, inv_sx, True)
        yi, fy = _area_linear_coords(h, height, scale_y, inv_sy, False)
        xi1, yi1 = np.minimum(xi + 1, w - 1), np.minimum(yi + 1, h - 1)
        a0 = (np.float32(1.0) - fx).astype(wt)[None, :, None]
        a1 = fx.astype(wt)[None, :, None]
        rows = s[:, xi] * a0 + s[:, xi1] * a1
        b0 = (np.float32(1.0) - fy).astype(wt)[:, None, None]
        b1 = fy.astype(wt)[:, None, None]
        out = rows[yi] * b0 + rows[yi1] * b1
    if is_u8:
        return np.clip(np.rint(out), 0, 255).astype(np.uint8)   # saturate_cast<uchar>(float): round half to even
    return out


# ------------------------------------------------------------------------------------------------------------------------
# networks (fp32, torch CPU).  Input [n, A, B, 3] as the package feeds it (already transposed), outputs in Keras order.
# ------------------------------------------------------------------------------------------------------------------------
def _same_pool(x, k, s):
    # Keras 'same': out = ceil(in / s), padding (never counted) split before = total // 2
    n, c, hh, ww = x.shape
    oh, ow = -(-hh // s), -(-ww // s)
    ph, pw = max((oh - 1) * s + k - hh, 0), max((ow - 1) * s + k - ww, 0)
    x = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), value=float("-inf"))
    return F.max_pool2d(x, k, s)


def run_net(net: str, weights: dict, x: np.ndarray):
    """-> list of head outputs (numpy float32) BEFORE reordering: [class probabilities, box offsets, (landmarks)]."""
    spec = NETS[net]
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).permute(0, 3, 1, 2)
    flat = False

    def apply(item, t, flat):
        if item[0] == "conv":
            k = torch.from_numpy(weights[f"{net}/{item[1]}/kernel"]).permute(3, 2, 0, 1).contiguous()
            return F.conv2d(t, k, torch.from_numpy(weights[f"{net}/{item[1]}/bias"])), flat
        if item[0] == "prelu":
            a = torch.from_numpy(weights[f"{net}/{item[1]}/alpha"]).reshape(-1)
            a = a.view(1, -1) if flat else a.view(1, -1, 1, 1)
            return torch.where(t > 0, t, a * t), flat
        if item[0] == "pool":
            _, k, s, mode = item
            return (_same_pool(t, k, s) if mode == "same" else F.max_pool2d(t, k, s)), flat
        if item[0] == "dense":
            if not flat:
                t = t.permute(0, 2, 3, 1).reshape(t.shape[0], -1)   # Keras Flatten of an NHWC tensor
            return t @ torch.from_numpy(weights[f"{net}/{item[1]}/kernel"]) + torch.from_numpy(weights[f"{net}/{item[1]}/bias"]), True
        raise ValueError(item)

    with torch.no_grad():
        for item in spec["body"]:
            t, flat = apply(item, t, flat)
        heads = []
        for item in spec["heads"]:
            o, f2 = apply(item, t, flat)
            heads.append(o if f2 else o.permute(0, 2, 3, 1))
        logits = heads[0]
        m = logits.max(dim=-1, keepdim=True).values
        e = torch.exp(logits - m)
        heads[0] = e / e.sum(dim=-1, keepdim=True)
    return [o.numpy() for o in heads]


# ------------------------------------------------------------------------------------------------------------------------
# box arithmetic
# ------------------------------------------------------------------------------------------------------------------------
def scale_pyramid(height, width, min_face_size=20, factor=0.709):
    m = 12 / min_face_size
    min_layer = np.amin([height, width]) * m
    scales, count = [], 0
    while min_layer >= 12:
        scales.append(m * np.power(factor, count))
        min_layer = min_layer * factor
        count += 1
    return scales


def generate_bounding_box(imap, reg, scale, t):
    """imap [H', W'] face probability, reg [H', W', 4] -> rows (x1, y1, x2, y2, score, dx1, dy1, dx2, dy2)."""
    stride, cellsize = 2, 12
    imap = imap.T
    planes = [reg[:, :, i].T for i in range(4)]
    a, b = np.where(imap >= t)          # a runs over image x, b over image y
    if a.shape[0] == 1:                 # the package flips the offset planes when exactly one cell fires
        planes = [np.flipud(p) for p in planes]
    score = imap[(a, b)]
    off = np.stack([p[(a, b)] for p in planes], axis=1) if a.size else np.empty((0, 4), np.float32)
    bb = np.stack([a, b], axis=1)
    q1 = np.fix((stride * bb + 1) / scale)
    q2 = np.fix((stride * bb + cellsize) / scale)
    return np.hstack([q1, q2, score[:, None], off])


def nms(boxes, threshold, method):
    if boxes.size == 0:
        return np.empty((0,), np.int64)
    x1, y1, x2, y2, s = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3], boxes[:, 4]
    area = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = np.argsort(s)
    pick = []
    while order.size > 0:
        i = order[-1]
        pick.append(i)
        idx = order[:-1]
        w = np.maximum(0.0, np.minimum(x2[i], x2[idx]) - np.maximum(x1[i], x1[idx]) + 1)
        h = np.maximum(0.0, np.minimum(y2[i], y2[idx]) - np.maximum(y1[i], y1[idx]) + 1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            if method == "Min":
                o = inter / np.minimum(area[i], area[idx])
            else:
                o = inter / (area[i] + area[idx] - inter)
        order = order[np.where(o <= threshold)]
    return np.asarray(pick, np.int64)


def rerec(b):
    h, w = b[:, 3] - b[:, 1], b[:, 2] - b[:, 0]
    side = np.maximum(w, h)
    b[:, 0] = b[:, 0] + w * 0.5 - side * 0.5
    b[:, 1] = b[:, 1] + h * 0.5 - side * 0.5
    b[:, 2:4] = b[:, 0:2] + np.transpose(np.tile(side, (2, 1)))
    return b


def bbreg(b, reg):
    w = b[:, 2] - b[:, 0] + 1
    h = b[:, 3] - b[:, 1] + 1
    b[:, 0:4] = np.transpose(np.vstack([b[:, 0] + reg[:, 0] * w, b[:, 1] + reg[:, 1] * h, b[:, 2] + reg[:, 2] * w, b[:, 3] + reg[:, 3] * h]))
    return b


def crop_windows(boxes, width, height):
    """The package's __pad, stated as what it does: box k is the window of tmpw x tmph pixels whose first pixel is frame pixel
    (x-1, y-1) (1-based box corners), everything outside the frame zero.  -> int32 [n, 4] = (ox, oy, cw, ch)."""
    x1 = boxes[:, 0].astype(np.int32)
    y1 = boxes[:, 1].astype(np.int32)
    cw = (boxes[:, 2] - boxes[:, 0] + 1).astype(np.int32)
    ch = (boxes[:, 3] - boxes[:, 1] + 1).astype(np.int32)
    return np.stack([x1 - 1, y1 - 1, cw, ch], axis=1).astype(np.int32)


def crop_resize(img, windows, size):
    """-> float64 [n, size(y), size(x), 3]: zero-padded crop, INTER_AREA resize, no rounding."""
    H, W = img.shape[:2]
    out = np.zeros((len(windows), size, size, 3))
    for k, (ox, oy, cw, ch) in enumerate(windows):
        if cw <= 0 or ch <= 0:
            continue
        tmp = np.zeros((ch, cw, 3))
        ys, ye, xs, xe = max(oy, 0), min(oy + ch, H), max(ox, 0), min(ox + cw, W)
        if ye > ys and xe > xs:
            tmp[ys - oy:ye - oy, xs - ox:xe - ox] = img[ys:ye, xs:xe]
        out[k] = resize_area(tmp, size, size)
    return out


class Nets:
    """The three networks as callables; tests substitute the GPU ones to compare the box logic exactly."""

    def __init__(self, weights):
        self.weights = weights

    def pnet(self, x):
        prob, reg = run_net("pnet", self.weights, x)
        return reg, prob                      # Keras model output order: [offsets, probabilities]

    def rnet(self, x):
        prob, reg = run_net("rnet", self.weights, x)
        return reg, prob

    def onet(self, x):
        prob, reg, pts = run_net("onet", self.weights, x)
        return reg, pts, prob


def stage1_input(img, scale):
    h, w = img.shape[:2]
    ws, hs = int(np.ceil(w * scale)), int(np.ceil(h * scale))
    return (resize_area(img, ws, hs) - 127.5) * 0.0078125


def detect_faces(img: np.ndarray, nets, min_face_size=20, steps_threshold=(0.6, 0.7, 0.7), scale_factor=0.709, trace=None):
    """img uint8 [H, W, 3] -> list of {'box': [x, y, w, h], 'confidence': p, 'keypoints': {...}} (package's output format)."""
    height, width = img.shape[:2]
    total = np.empty((0, 9))
    for scale in scale_pyramid(height, width, min_face_size, scale_factor):
        x = stage1_input(img, scale)
        reg, prob = nets.pnet(np.transpose(x[None], (0, 2, 1, 3)))
        reg, prob = np.transpose(reg, (0, 2, 1, 3)), np.transpose(prob, (0, 2, 1, 3))
        boxes = generate_bounding_box(prob[0, :, :, 1].copy(), reg[0].copy(), scale, steps_threshold[0])
        pick = nms(boxes.copy(), 0.5, "Union")
        if boxes.size > 0 and pick.size > 0:
            total = np.append(total, boxes[pick, :], axis=0)
    if trace is not None:
        trace["stage1_raw"] = total.copy()
    points = np.empty((0,))
    if total.shape[0] > 0:
        total = total[nms(total.copy(), 0.7, "Union"), :]
        rw, rh = total[:, 2] - total[:, 0], total[:, 3] - total[:, 1]
        total = np.transpose(np.vstack([total[:, 0] + total[:, 5] * rw, total[:, 1] + total[:, 6] * rh,
                                        total[:, 2] + total[:, 7] * rw, total[:, 3] + total[:, 8] * rh, total[:, 4]]))
        total = rerec(total.copy())
        total[:, 0:4] = np.fix(total[:, 0:4]).astype(np.int32)
    if trace is not None:
        trace["stage1"] = total.copy()
    if total.shape[0] > 0:   # stage 2
        crops = (crop_resize(img, crop_windows(total, width, height), 24) - 127.5) * 0.0078125
        reg, prob = nets.rnet(np.transpose(crops, (0, 2, 1, 3)))
        score = prob[:, 1]
        ok = np.where(score > steps_threshold[1])[0]
        total = np.hstack([total[ok, 0:4].copy(), score[ok, None].copy()])
        mv = reg[ok]
        if total.shape[0] > 0:
            pick = nms(total, 0.7, "Union")
            total = rerec(bbreg(total[pick, :].copy(), mv[pick]).copy())
    if trace is not None:
        trace["stage2"] = total.copy()
    if total.shape[0] > 0:   # stage 3
        total = np.fix(total).astype(np.int32)
        crops = (crop_resize(img, crop_windows(total, width, height), 48) - 127.5) * 0.0078125
        reg, pts, prob = nets.onet(np.transpose(crops, (0, 2, 1, 3)))
        score = prob[:, 1]
        ok = np.where(score > steps_threshold[2])[0]
        points = np.array(pts.T[:, ok])
        total = np.hstack([total[ok, 0:4].copy(), score[ok, None].copy()])
        mv = reg[ok]
        w = total[:, 2] - total[:, 0] + 1
        h = total[:, 3] - total[:, 1] + 1
        points[0:5, :] = np.tile(w, (5, 1)) * points[0:5, :] + np.tile(total[:, 0], (5, 1)) - 1
        points[5:10, :] = np.tile(h, (5, 1)) * points[5:10, :] + np.tile(total[:, 1], (5, 1)) - 1
        if total.shape[0] > 0:
            total = bbreg(total.copy(), mv)
            pick = nms(total.copy(), 0.7, "Min")
            total, points = total[pick, :], points[:, pick]
    else:
        total = np.empty((0, 5))
    if trace is not None:
        trace["stage3"] = total.copy()
        trace["points"] = np.array(points)
    faces = []
    for box, kp in zip(total, points.T if total.shape[0] else []):
        x, y = max(0, int(box[0])), max(0, int(box[1]))
        faces.append({"box": [x, y, int(box[2] - x), int(box[3] - y)], "confidence": box[-1],
                      "keypoints": {"left_eye": (int(kp[0]), int(kp[5])), "right_eye": (int(kp[1]), int(kp[6])),
                                    "nose": (int(kp[2]), int(kp[7])), "mouth_left": (int(kp[3]), int(kp[8])),
                                    "mouth_right": (int(kp[4]), int(kp[9]))}})
    return faces
