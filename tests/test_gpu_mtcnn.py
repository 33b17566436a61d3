"""MTCNN face detector (BASELINE config 5) through the C ABI against oracle/mtcnn_oracle.py.
PARITY UNPINNED: the reference wraps the PyPI `mtcnn` package (detectors/face_detector.py:63-78), which -- like OpenCV -- is not
installed here, and ships no detector fixtures; the oracle restates the published algorithm.  Weights are synthetic.
Bars: byte / index work (resampling, crops, pooling, cell compaction, the whole box logic on identical network outputs) is
bit-exact; the f16 networks are compared with the fp32 oracle inside a stated tolerance."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from facenet_amd import _lib
from facenet_amd.detectors import mtcnn as gm
from facenet_amd.detectors.face_detector import BoundingBox, FaceDetector
from oracle import mtcnn_oracle as mo
from tests.util import conv_desc, ptr, rel_err, stream

pytestmark = pytest.mark.gpu
HF = _lib.FN_F16
FACE_BIAS = (0.5, 1.0, 1.0)   # shifts the face logits of the synthetic networks so that every stage passes some candidates


def _frame(h, w, seed=0, cell=8):
    """Blocky random image + noise: smooth enough for stable network responses, textured enough to exercise the resampling."""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (-(-h // cell), -(-w // cell), 3), dtype=np.uint8)
    img = np.kron(base, np.ones((cell, cell, 1), np.uint8))[:h, :w].astype(np.int32) + rng.integers(-12, 13, (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.fixture(scope="module")
def weights():
    return mo.random_weights(0, face_bias=FACE_BIAS)


@pytest.fixture(scope="module")
def detector(weights):
    return gm.MTCNN(weights=weights)


def _resize_gpu(lib, frame, windows, oh, ow, u8):
    ft = torch.from_numpy(frame).cuda()
    win = torch.from_numpy(np.asarray(windows, np.int32)).cuda()
    n = win.shape[0]
    out = torch.full((n, ow, oh, 8), 7.0, dtype=torch.float16, device="cuda")
    _lib.check(lib.fn_area_resize_crop(ptr(ft), frame.shape[0], frame.shape[1], ptr(win), n, oh, ow, 1 if u8 else 0, ptr(out), HF, stream()))
    return out.cpu().numpy()


@pytest.mark.parametrize("hw,scale", [((120, 160), 0.6), ((120, 160), 0.6 * 0.709 ** 3), ((97, 131), 0.3017), ((720, 1280), 0.6 * 0.709 ** 2),
                                      ((64, 64), 0.5), ((50, 70), 1.0)])
def test_pyramid_level_is_bit_exact(lib, hw, scale):
    """uint8 frame -> INTER_AREA -> uint8 (round half even) -> (v - 127.5) / 128, transposed: every value is exact in f16."""
    img = _frame(*hw, seed=3)
    hs, ws = int(np.ceil(hw[0] * scale)), int(np.ceil(hw[1] * scale))
    got = _resize_gpu(lib, img, [[0, 0, hw[1], hw[0]]], hs, ws, True)
    ref = mo.stage1_input(img, scale)                               # [hs, ws, 3] float64
    assert got.shape == (1, ws, hs, 8)
    assert np.array_equal(got[0, :, :, :3].astype(np.float64), np.transpose(ref, (1, 0, 2)))
    assert not got[..., 3:].any()
    # the two-pass whole-frame form the detector uses: same bits
    ft = torch.from_numpy(img).cuda()
    rows = torch.empty(hw[0] * ws * 3, dtype=torch.float32, device="cuda")
    out = torch.full((1, ws, hs, 8), 7.0, dtype=torch.float16, device="cuda")
    _lib.check(lib.fn_area_resize_frame(ptr(ft), hw[0], hw[1], hs, ws, ptr(rows), ptr(out), HF, stream()))
    assert np.array_equal(out.cpu().numpy(), got)
    with pytest.raises(ValueError):
        _lib.check(lib.fn_area_resize_frame(ptr(ft), hw[0], hw[1], hw[0] + 1, ws, ptr(rows), ptr(out), HF, stream()))


@pytest.mark.parametrize("size", [24, 48])
def test_candidate_crops_are_bit_exact(lib, size):
    """float64 crops: windows inside, across every border, fully outside, smaller than the target (enlarging path), ragged
    aspect (one axis shrinks, one grows), 1-pixel and empty windows.  GPU = double arithmetic -> float -> f16, same as the
    oracle's float64 -> float32 (what Keras feeds) -> f16 (the storage type)."""
    img = _frame(90, 130, seed=5)
    windows = [[10, 12, 60, 60], [-7, -9, 40, 40], [100, 60, 50, 50], [-30, 20, 200, 45], [3, 4, 11, 11], [50, 50, 24, 24], [20, 30, size, size],
               [5, 70, 100, 13], [60, 2, 9, 80], [40, 40, 1, 1], [200, 200, 30, 30], [0, 0, 130, 90], [17, 23, 0, 0], [88, 61, 47, 53]]
    got = _resize_gpu(lib, img, windows, size, size, False)
    ref = (mo.crop_resize(img, np.asarray(windows, np.int32), size) - 127.5) * 0.0078125     # [n, y, x, 3]
    ref16 = np.transpose(ref, (0, 2, 1, 3)).astype(np.float32).astype(np.float16)
    assert np.array_equal(got[..., :3], ref16)
    assert not got[..., 3:].any()


@pytest.mark.parametrize("k,stride,same", [(2, 2, True), (3, 2, True), (3, 2, False), (2, 2, False)])
@pytest.mark.parametrize("shape", [(3, 22, 22, 32), (1, 215, 383, 16), (2, 21, 10, 64), (5, 8, 8, 64)])
def test_maxpool2d(lib, k, stride, same, shape):
    n, a, b, c = shape
    x = torch.randn(shape, generator=torch.Generator().manual_seed(1)).to(torch.float16).cuda()
    oa, ob = (-(-a // stride), -(-b // stride)) if same else ((a - k) // stride + 1, (b - k) // stride + 1)
    pa = max((oa - 1) * stride + k - a, 0) // 2 if same else 0      # Keras 'same': smaller half of the padding in front
    pb = max((ob - 1) * stride + k - b, 0) // 2 if same else 0
    y = torch.zeros(n, oa, ob, c, dtype=torch.float16, device="cuda")
    _lib.check(lib.fn_maxpool2d_fwd(ptr(x), c, ptr(y), c, n, a, b, c, k, stride, pa, pb, oa, ob, HF, stream()))
    xr = x.float().cpu().permute(0, 3, 1, 2)
    ref = mo._same_pool(xr, k, stride) if same else F.max_pool2d(xr, k, stride)
    assert torch.equal(y.float().cpu(), ref.permute(0, 2, 3, 1))
    with pytest.raises(ValueError):
        _lib.check(lib.fn_maxpool2d_fwd(ptr(x), c, ptr(y), c, n, a, b, c, k, stride, 0, 0, oa + 3, ob, HF, stream()))


@pytest.mark.parametrize("case", [(1, 130, 90, 8, 16, 3, 3), (1, 64, 43, 16, 16, 3, 3), (7, 11, 11, 32, 48, 3, 3), (9, 4, 4, 48, 64, 2, 2), (9, 3, 3, 64, 128, 3, 3)])
@pytest.mark.parametrize("tile", [0, 64032])
def test_conv_bias_prelu_epilogue(lib, case, tile):
    """fn_conv_desc.prelu: y = v > 0 ? v : slope[c] * v on conv + bias, through the library's own kernel choice (halo-tile kernel
    for the large maps) and through the implicit-GEMM kernel."""
    n, a, b, ci, co, kh, kw = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, a, b, ci, generator=g).to(torch.float16).cuda()
    w = (torch.randn(co, kh, kw, ci, generator=g) * 0.2).to(torch.float16).cuda()
    bias = (torch.randn(co, generator=g) * 0.3).cuda()
    slope = (torch.rand(co, generator=g) * 0.5).cuda()
    d = conv_desc(n, a, b, ci, co, kh, kw, 1, 0, 0, HF)
    y = torch.zeros(n, d.OH, d.OW, co, dtype=torch.float16, device="cuda")
    d.x, d.w, d.y, d.bias, d.prelu, d.tile_fwd = ptr(x), ptr(w), ptr(y), ptr(bias), ptr(slope), tile
    _lib.check(lib.fn_conv2d_fwd(C.byref(d), stream()))
    v = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.float().cpu().permute(0, 3, 1, 2), bias.cpu())
    ref = torch.where(v > 0, v, slope.cpu().view(1, -1, 1, 1) * v).permute(0, 2, 3, 1)
    assert rel_err(y, ref) < 1e-3
    d.relu = 1
    with pytest.raises(ValueError):
        _lib.check(lib.fn_conv2d_fwd(C.byref(d), stream()))


@pytest.mark.parametrize("n", [17, 64, 65, 129, 1000, 6000])
@pytest.mark.parametrize("thr,by_min", [(0.5, False), (0.7, False), (0.7, True)])
def test_device_nms_equals_the_package_loop(n, thr, by_min):
    """fn_nms_greedy == the oracle's greedy loop, index for index: clustered boxes, tied scores (the order is the host's
    np.argsort in both), degenerate boxes whose ratios are negative, infinite or NaN."""
    rng = np.random.default_rng(n)
    centres = rng.uniform(0, 400, (max(n // 12, 2), 2))
    xy = centres[rng.integers(0, len(centres), n)] + rng.normal(0, 6, (n, 2))
    wh = rng.uniform(15, 60, (n, 2))
    boxes = np.hstack([np.fix(xy), np.fix(xy + wh), rng.uniform(0.6, 1.0, (n, 1)).astype(np.float32).astype(np.float64), rng.normal(size=(n, 4))])
    boxes[rng.integers(0, n, n // 5), 4] = boxes[0, 4]                 # ties
    boxes[3, 2:4] = boxes[3, 0:2] - 1                                  # zero area
    boxes[5, 2:4] = boxes[5, 0:2] - 9                                  # negative extents
    ref = mo.nms(boxes.copy(), thr, "Min" if by_min else "Union")
    got = gm._DeviceNms(torch.device("cuda:0")).run([(boxes, thr, by_min), (boxes[: n // 2], thr, by_min)])
    assert np.array_equal(got[0], ref)
    assert np.array_equal(got[1], mo.nms(boxes[: n // 2].copy(), thr, "Min" if by_min else "Union"))
    assert np.array_equal(gm._iou_keep(boxes, thr, by_min), ref)


def _gpu_net(det, net, x):
    """Run one product network on the oracle's input x [n, A, B, 3] (float64, already normalised and transposed)."""
    lib = _lib.load()
    network = det._nets[net]
    n, a, b = x.shape[:3]
    cap = 64
    while cap < n:
        cap *= 2
    plan = network._plan("test", cap, a, b)
    plan["x"].zero_()
    plan["x"][:n, :, :, :3] = torch.from_numpy(np.ascontiguousarray(x)).to(torch.float32).to(torch.float16).cuda()
    out = network.run(plan, n, stream())
    torch.cuda.synchronize()
    return out[:n]


class GpuNets:
    """The oracle's cascade driven by the product's networks: isolates the box logic (must then agree exactly)."""

    def __init__(self, det):
        self.det = det

    def pnet(self, x):
        lib = _lib.load()
        out = _gpu_net(self.det, "pnet", x)
        _, a, b, ld = out.shape
        cand = torch.empty(a * b, 8, dtype=torch.float32, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        _lib.check(lib.fn_mtcnn_candidates(ptr(out), a * b, ld, -1.0, ptr(cand), ptr(cnt), a * b, 0, 1, stream()))
        assert int(cnt.item()) == a * b
        c = cand.cpu().numpy()
        c = c[np.argsort(c[:, 0].copy().view(np.int32))]
        prob1 = c[:, 2].reshape(1, a, b)
        return c[:, 3:7].reshape(1, a, b, 4), np.stack([1 - prob1, prob1], axis=-1)

    def rnet(self, x):
        rows = _gpu_net(self.det, "rnet", x).reshape(x.shape[0], -1).cpu().numpy()
        return rows[:, 2:6], gm._softmax2(rows[:, 0:2])

    def onet(self, x):
        rows = _gpu_net(self.det, "onet", x).reshape(x.shape[0], -1).cpu().numpy()
        return rows[:, 2:6], rows[:, 6:16], gm._softmax2(rows[:, 0:2])


def test_networks_against_the_fp32_oracle(detector, weights):
    """f16 storage, fp32 accumulation vs fp32 torch: outputs within 2e-2 relative (L2) and 0.02 absolute on probabilities."""
    rng = np.random.default_rng(2)
    nets = mo.Nets(weights)
    gn = GpuNets(detector)
    xp = (rng.integers(0, 256, (1, 77, 53, 3)) - 127.5) * 0.0078125
    (reg_o, prob_o), (reg_g, prob_g) = nets.pnet(xp), gn.pnet(xp)
    assert reg_o.shape == reg_g.shape == (1, 34, 22, 4)       # 77 -> 75 -> 38 -> 36 -> 34
    assert rel_err(torch.from_numpy(reg_g), torch.from_numpy(reg_o)) < 2e-2
    assert np.abs(prob_g[..., 1] - prob_o[..., 1]).max() < 0.02
    for name, size in (("rnet", 24), ("onet", 48)):
        x = (rng.integers(0, 256, (37, size, size, 3)) - 127.5) * 0.0078125
        o, g = getattr(nets, name)(x), getattr(gn, name)(x)
        for oo, gg in zip(o[:-1], g[:-1]):
            assert oo.shape == gg.shape and rel_err(torch.from_numpy(gg), torch.from_numpy(oo)) < 2e-2
        assert np.abs(g[-1][:, 1] - o[-1][:, 1]).max() < 0.02


def test_candidate_compaction_is_complete(lib, detector):
    """Every cell with p >= threshold exactly once, with its own score and offsets; the counter reports overflow."""
    frame = torch.from_numpy(_frame(240, 320, seed=9)).cuda()
    reg, prob = detector.pnet_maps(frame, 0)
    a, b = prob.shape
    out = torch.zeros(a * b, 8, dtype=torch.float32, device="cuda")
    logit = np.log(np.clip(prob, 1e-6, 1 - 1e-6) / np.clip(1 - prob, 1e-6, 1))
    out[:, 1] = torch.from_numpy(logit.reshape(-1)).cuda()
    out[:, 2:6] = torch.from_numpy(reg.reshape(-1, 4)).cuda()
    cand = torch.zeros(a * b, 8, dtype=torch.float32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(lib.fn_mtcnn_candidates(ptr(out), a * b, 8, 0.6, ptr(cand), ptr(cnt), a * b, 5, 1, stream()))
    n = int(cnt.item())
    c = cand[:n].cpu().numpy()
    cells = np.sort(c[:, 0].copy().view(np.int32))
    l = out[:, :2].cpu().numpy()
    e = np.exp(l - l.max(1, keepdims=True))
    p1 = e[:, 1] / e.sum(1)
    margin = np.abs(p1 - 0.6) > 1e-6
    expect = np.where(p1 >= 0.6)[0]
    assert 0 < n < a * b and len(np.unique(cells)) == n and np.all(c[:, 1].copy().view(np.int32) == 5)
    assert np.array_equal(np.intersect1d(cells, np.where(margin)[0]), np.intersect1d(expect, np.where(margin)[0]))
    row = c[0]
    cell = int(row[0:1].view(np.int32)[0])
    assert np.array_equal(row[3:7], out[cell, 2:6].cpu().numpy()) and abs(row[2] - p1[cell]) < 1e-6 and row[7] == 0
    _lib.check(lib.fn_mtcnn_candidates(ptr(out), a * b, 8, 0.6, ptr(cand), ptr(cnt), 5, 5, 0, stream()))   # room for 5 only, counter carries on
    assert int(cnt.item()) == 2 * n


@pytest.mark.parametrize("hw,seed", [((120, 160), 1), ((201, 143), 2), ((96, 256), 3)])
def test_cascade_box_logic_is_exact_on_identical_network_outputs(detector, hw, seed):
    """Product cascade == oracle cascade when the oracle is handed the product's network outputs: pyramid, crops, cell order,
    box generation, NMS, squares, regression, landmark mapping and the result dictionaries, bit for bit."""
    img = _frame(*hw, seed=seed)
    to, tg = {}, {}
    faces_o = mo.detect_faces(img, GpuNets(detector), trace=to)
    total, points = detector.detect_boxes(img, trace=tg)
    assert to["stage1"].shape[0] > 20 and to["stage2"].shape[0] > 5 and to["stage3"].shape[0] > 0
    for k in ("stage1", "stage2", "stage3"):
        assert np.array_equal(to[k], tg[k]), k
    assert np.array_equal(to["points"], points) and points.dtype == np.float32
    faces_g = detector.detect_faces(img)
    assert faces_g == faces_o and set(faces_g[0]) == {"box", "confidence", "keypoints"}
    assert set(faces_g[0]["keypoints"]) == {"left_eye", "right_eye", "nose", "mouth_left", "mouth_right"}


def _iou(a, b):
    iw = max(0.0, min(a[2], b[2]) - max(a[0], b[0]) + 1)
    ih = max(0.0, min(a[3], b[3]) - max(a[1], b[1]) + 1)
    inter = iw * ih
    return inter / ((a[2] - a[0] + 1) * (a[3] - a[1] + 1) + (b[2] - b[0] + 1) * (b[3] - b[1] + 1) - inter)


def test_end_to_end_against_the_fp32_oracle(detector, weights):
    """f16 networks vs the fp32 oracle end to end.  Decisions next to a threshold may flip, so the bar is on the stage-1 boxes
    (>= 97 % of the oracle's have an identical product box) and on the final faces (>= 80 % matched at IoU > 0.7 with the
    confidence within 0.03) -- measured: see DESIGN.md section 9."""
    img = _frame(160, 200, seed=4)
    to, tg = {}, {}
    mo.detect_faces(img, mo.Nets(weights), trace=to)
    detector.detect_boxes(img, trace=tg)
    s1o = {tuple(r[:4]) for r in to["stage1"]}
    s1g = {tuple(r[:4]) for r in tg["stage1"]}
    assert len(s1o) > 50 and len(s1o & s1g) >= 0.97 * len(s1o) and len(s1o & s1g) >= 0.97 * len(s1g)
    fo_, fg = to["stage3"], tg["stage3"]
    assert fo_.shape[0] > 3
    hits = sum(any(_iou(o, g) > 0.7 and abs(o[4] - g[4]) < 0.03 for g in fg) for o in fo_)
    assert hits >= 0.8 * fo_.shape[0] and abs(fg.shape[0] - fo_.shape[0]) <= max(2, 0.2 * fo_.shape[0])


def test_hd_frame_properties():
    """BASELINE config 5 size (1280x720): eleven pyramid levels, deterministic replay, squares out of stage 1, an idempotent final
    NMS, and a candidate buffer that is too small at first (max_candidates=256: the overflowing levels are re-run)."""
    detector = gm.MTCNN(weights=mo.random_weights(0, face_bias=(-0.3, 1.0, 1.0)), max_candidates=256)
    img = _frame(720, 1280, seed=6, cell=24)
    assert len(detector._scales(720, 1280)) == len(mo.scale_pyramid(720, 1280)) == 11
    t1, t2 = {}, {}
    a, pa = detector.detect_boxes(img, trace=t1)
    b, pb = detector.detect_boxes(torch.from_numpy(img).cuda(), trace=t2)
    assert a.shape[0] > 0 and np.array_equal(a, b) and np.array_equal(pa, pb) and np.array_equal(t1["stage1"], t2["stage1"])
    assert np.all(a[:, 4] > 0.7) and t1["stage1"].shape[0] > 256
    big = gm.MTCNN(weights=mo.random_weights(0, face_bias=(-0.3, 1.0, 1.0)))          # default buffer: no re-run
    c, pc = big.detect_boxes(img)
    assert np.array_equal(a, c) and np.array_equal(pa, pc)
    s1 = t1["stage1"]
    assert np.array_equal(s1[:, :4], np.fix(s1[:, :4])) and np.allclose(s1[:, 2] - s1[:, 0], s1[:, 3] - s1[:, 1], atol=3)   # squares up to the truncation toward zero
    keep = gm._iou_keep(a.copy(), 0.7, True)
    assert len(keep) == a.shape[0]                                   # idempotent: the final NMS has nothing left to remove


def test_api_errors_and_wrappers(detector, weights, tmp_path):
    with pytest.raises(gm.InvalidImage):
        detector.detect_faces(None)
    with pytest.raises(gm.InvalidImage):
        detector.detect_faces(np.zeros((10, 10, 3), np.float32))
    with pytest.raises(ValueError):
        gm.MTCNN()
    with pytest.raises(ValueError):
        FaceDetector(detector="nope")
    assert detector.detect_faces(np.zeros((11, 30, 3), np.uint8)) == []          # smaller than one 12x12 cell at every scale
    path = tmp_path / "mtcnn.npz"
    np.savez(path, **weights)
    fd = FaceDetector(detector="pypimtcnn", weights_file=str(path))
    img = _frame(120, 160, seed=1)
    boxes = fd.detect(img)
    faces = detector.detect_faces(img)
    assert fd.mode == "RGB" and len(boxes) == len(faces) > 0 and isinstance(boxes[0], BoundingBox)
    assert [boxes[0].left, boxes[0].top, boxes[0].width, boxes[0].height] == faces[0]["box"]
    lists = {net: [weights[k] for k, _ in mo.variable_shapes(net)] for net in ("pnet", "rnet", "onet")}
    back = gm.weights_from_lists(lists)
    assert set(back) == set(weights) and all(np.array_equal(back[k], weights[k]) for k in weights)


def test_extract_faces_app(detector, weights, tmp_path):
    """apps/extract_faces.py:37-80: thumbnails per class, `_n` suffix for further faces, single-face filter, unreadable files counted."""
    from types import SimpleNamespace

    from PIL import Image

    from facenet_amd.apps.extract_faces import extract_faces
    src = tmp_path / "in" / "alice"
    src.mkdir(parents=True)
    Image.fromarray(_frame(120, 160, seed=1)).save(src / "a.png")
    Image.fromarray(np.zeros((11, 30, 3), np.uint8)).save(src / "tiny.png")     # no pyramid level: no face
    (src / "broken.jpg").write_bytes(b"not an image")
    path = tmp_path / "w.npz"
    np.savez(path, **weights)
    fd = FaceDetector(detector="pypimtcnn", weights_file=str(path))
    cls = SimpleNamespace(name="alice", files=sorted(str(p) for p in src.iterdir()))
    opts = SimpleNamespace(size=160, margin=0.25)
    out = tmp_path / "out"
    stats = extract_faces([cls], out, fd, opts, detect_multiple_faces=True, log=lambda *a: None)
    n = len(detector.detect_faces(_frame(120, 160, seed=1)))
    assert n > 1 and stats["extracted"] == 1 and stats["unread"] == 1 and len(stats["sizes"]) == n
    names = sorted(p.name for p in (out / "alice").iterdir())
    assert "a.png" in names and f"a_{n - 1}.png" in names and len(names) == n
    assert Image.open(out / "alice" / "a.png").size == (200, 200) and (out / "sizes.json").exists()
    single = extract_faces([cls], tmp_path / "out1", fd, opts, detect_multiple_faces=False, log=lambda *a: None)
    assert single["extracted"] == 0 and not list((tmp_path / "out1" / "alice").iterdir())


def test_detector_entry_points_reject_bad_arguments(lib):
    """C-ABI error behaviour: rc = FN_EINVAL (-> ValueError) with a message, nothing launched."""
    z8 = torch.zeros(64, dtype=torch.uint8, device="cuda")
    zf = torch.zeros(64, dtype=torch.float32, device="cuda")
    zi = torch.zeros(64, dtype=torch.int32, device="cuda")
    zd = torch.zeros(64, dtype=torch.float64, device="cuda")
    with pytest.raises(ValueError):       # no boxes
        _lib.check(lib.fn_area_resize_crop(ptr(z8), 4, 4, ptr(zi), 0, 24, 24, 0, ptr(zf), HF, stream()))
    with pytest.raises(ValueError):       # unknown dtype
        _lib.check(lib.fn_area_resize_crop(ptr(z8), 4, 4, ptr(zi), 1, 24, 24, 0, ptr(zf), 7, stream()))
    with pytest.raises(ValueError):       # enlarging a uint8 frame
        _lib.check(lib.fn_area_resize_frame(ptr(z8), 4, 4, 8, 4, ptr(zf), ptr(zf), HF, stream()))
    with pytest.raises(ValueError):       # channels not a multiple of 8
        _lib.check(lib.fn_maxpool2d_fwd(ptr(zf), 12, ptr(zf), 12, 1, 2, 2, 12, 2, 2, 0, 0, 1, 1, HF, stream()))
    with pytest.raises(ValueError):       # record buffer not 16-byte aligned
        _lib.check(lib.fn_mtcnn_candidates(ptr(zf), 4, 8, 0.5, ptr(zf) + 4, ptr(zi), 4, 0, 1, stream()))
    with pytest.raises(ValueError):       # bit-matrix workspace too small: 100 boxes need 100 * 2 * 8 bytes
        _lib.check(lib.fn_nms_greedy(ptr(zd), 5, ptr(zi), 100, 0.5, 0, ptr(zf), 64, ptr(zi), ptr(zi), stream()))
    assert b"workspace" in lib.fn_last_error()
