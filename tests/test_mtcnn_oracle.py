"""CPU checks of oracle/mtcnn_oracle.py (parity unpinned: neither the PyPI `mtcnn` package nor OpenCV is installed, the
reference ships no detector fixtures) and of the host-side mirrors in facenet_amd/detectors: properties the published algorithm
must have, the committed golden run, and oracle == product host logic on hand-made inputs."""
import os

import numpy as np
import pytest

from facenet_amd.detectors import mtcnn as gm
from facenet_amd.detectors.face_detector import BoundingBox, FaceDetector, image_processing
from oracle import mtcnn_oracle as mo

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "mtcnn_oracle_run.npz")


def test_area_resize_properties():
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (12, 18, 3), dtype=np.uint8)
    # integer ratios: the block mean (cv2's fast path computes the same average)
    m = a.reshape(4, 3, 6, 3, 3).astype(np.float64).mean(axis=(1, 3))
    assert np.abs(mo.resize_area(a, 6, 4).astype(np.float64) - m).max() <= 0.5
    assert np.allclose(mo.resize_area(a.astype(np.float64), 6, 4), m, atol=1e-4)
    # identity, constants, mass conservation for a non-integer ratio
    assert np.array_equal(mo.resize_area(a, 18, 12), a)
    c = np.full((13, 17, 3), 201, np.uint8)
    assert np.all(mo.resize_area(c, 7, 5) == 201) and np.allclose(mo.resize_area(c.astype(np.float64), 24, 24), 201)
    f = a.astype(np.float64)
    assert abs(mo.resize_area(f, 7, 5).mean() - f.mean()) < 1.0
    # enlargement: corners keep their values, range never exceeded
    up = mo.resize_area(f, 48, 48)
    assert up.shape == (48, 48, 3) and np.array_equal(up[0, 0], f[0, 0]) and up.min() >= f.min() and up.max() <= f.max()
    with pytest.raises(NotImplementedError):
        mo.resize_area(a, 40, 40)


def test_pyramid_and_network_shapes():
    s = mo.scale_pyramid(720, 1280)
    assert len(s) == 11 and abs(s[0] - 0.6) < 1e-12 and abs(s[1] / s[0] - 0.709) < 1e-12
    assert mo.scale_pyramid(11, 300) == []
    w = mo.random_weights(3)
    x = np.zeros((2, 24, 24, 3))
    prob, reg = mo.run_net("rnet", w, x)
    assert prob.shape == (2, 2) and reg.shape == (2, 4) and np.allclose(prob.sum(1), 1)
    prob, reg, pts = mo.run_net("onet", w, np.zeros((3, 48, 48, 3)))
    assert prob.shape == (3, 2) and reg.shape == (3, 4) and pts.shape == (3, 10)
    prob, reg = mo.run_net("pnet", w, np.zeros((1, 31, 12, 3)))
    assert prob.shape == (1, 11, 1, 2) and reg.shape == (1, 11, 1, 4)            # 31 -> 29 -> ceil(29 / 2) = 15 -> 13 -> 11
    n_vars = {net: len(mo.variable_shapes(net)) for net in mo.NETS}
    assert n_vars == {"pnet": 13, "rnet": 16, "onet": 21}


def test_box_logic_oracle_equals_product_host_code():
    rng = np.random.default_rng(5)
    xy = rng.uniform(0, 200, (300, 2))
    wh = rng.uniform(10, 80, (300, 2))
    boxes = np.hstack([xy, xy + wh, rng.uniform(0.6, 1.0, (300, 1))])
    for thr, method in ((0.5, "Union"), (0.7, "Union"), (0.7, "Min")):
        pick = mo.nms(boxes.copy(), thr, method)
        assert np.array_equal(pick, gm._iou_keep(boxes.copy(), thr, method == "Min"))
        assert 0 < len(pick) < 300 and pick[0] == np.argmax(boxes[:, 4])
        kept = boxes[pick]
        assert len(mo.nms(kept.copy(), thr, method)) == len(pick)                # idempotent
    assert mo.nms(np.empty((0, 5)), 0.5, "Union").size == 0
    assert np.array_equal(mo.rerec(boxes.copy()), gm._square(boxes.copy()))
    sq = mo.rerec(boxes.copy())
    assert np.allclose(sq[:, 2] - sq[:, 0], sq[:, 3] - sq[:, 1])
    reg = rng.normal(0, 0.1, (300, 4)).astype(np.float32)
    assert np.array_equal(mo.bbreg(boxes.copy(), reg), gm._regress(boxes.copy(), reg))
    ib = np.fix(boxes)
    assert np.array_equal(mo.crop_windows(ib, 300, 300), gm._windows(ib))
    # one firing cell: the offsets come from the x-flipped planes (package quirk), several cells: from their own position
    imap = np.zeros((5, 7), np.float32)
    reg4 = rng.normal(size=(5, 7, 4)).astype(np.float32)
    imap[2, 1] = 0.9
    one = mo.generate_bounding_box(imap, reg4, 0.5, 0.6)
    assert one.shape == (1, 9) and np.array_equal(one[0, :4], [np.fix(3 / 0.5), np.fix(5 / 0.5), np.fix(14 / 0.5), np.fix(16 / 0.5)])
    assert np.array_equal(one[0, 5:], reg4[2, 7 - 1 - 1])
    imap[4, 3] = 0.7
    two = mo.generate_bounding_box(imap, reg4, 0.5, 0.6)
    assert two.shape == (2, 9) and np.array_equal(two[0, 5:], reg4[2, 1]) and np.array_equal(two[1, 5:], reg4[4, 3])   # x-major order


def test_committed_golden_run():
    """The oracle today reproduces the run committed with oracle/make_golden.py (guards the restatement against drift)."""
    z = np.load(GOLDEN)
    w = mo.random_weights(int(z["weights_seed"]), face_bias=tuple(z["face_bias"]))
    tr = {}
    mo.detect_faces(z["image"], mo.Nets(w), trace=tr)
    assert np.array_equal(tr["stage1"][:, :4], z["stage1"][:, :4]) and np.allclose(tr["stage1"][:, 4], z["stage1"][:, 4], atol=1e-5)
    assert tr["stage3"].shape == z["stage3"].shape and np.allclose(tr["stage3"], z["stage3"], atol=1e-2)


def test_face_detector_mirror():
    b = BoundingBox(1.4, 2.6, 10, 20, 0.91234)
    assert (b.left, b.top, b.right, b.bottom, b.width, b.height) == (1, 3, 12, 24, 10, 20)
    assert b.left_upper == (1, 3) and b.right_lower == (12, 24) and b.confidence_as_string == "0.912"
    assert repr(b) == "left = 1, top = 3, width = 10, height = 20, confidence = 0.91234" and b.info() == "[1, 3, 10, 20, 0.91234]"
    from PIL import Image
    from types import SimpleNamespace
    img = Image.fromarray(np.random.default_rng(0).integers(0, 256, (60, 80, 3), dtype=np.uint8))
    out = image_processing(img, b, SimpleNamespace(margin=0.25, size=160))
    assert out.size == (200, 200)
    with pytest.raises(ValueError):
        image_processing(np.zeros((4, 4, 3)), b, SimpleNamespace(margin=0.25, size=160))
    with pytest.raises(ValueError):
        FaceDetector(detector="other")
    with pytest.raises(NotImplementedError):
        FaceDetector(detector="frcnnv3")


def test_weight_files_are_never_unpickled_by_the_product_path(tmp_path):
    """Advisor finding (round 2): `mtcnn.weights_file` reaches load_weights from the extract_faces app; only `.npz` with
    allow_pickle=False is accepted there.  The package's pickled mtcnn_weights.npy goes through the offline converter, whose
    restricted unpickler rebuilds arrays / lists / dicts and refuses every other global."""
    import os
    import subprocess
    import sys
    from facenet_amd.detectors import mtcnn as gm
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "convert_mtcnn_weights.py")
    w = mo.random_weights(seed=1)
    lists = {net: [w[k] for k, _ in mo.variable_shapes(net)] for net in ("pnet", "rnet", "onet")}
    np.save(tmp_path / "w.npy", lists, allow_pickle=True)
    with pytest.raises(ValueError):
        gm.load_weights(tmp_path / "w.npy")                       # the product path refuses the pickle outright
    assert subprocess.run([sys.executable, tool, str(tmp_path / "w.npy"), str(tmp_path / "w.npz")]).returncode != 0   # no opt-in
    subprocess.check_call([sys.executable, tool, "--i-trust-this-file", str(tmp_path / "w.npy"), str(tmp_path / "w.npz")])
    back = gm.load_weights(tmp_path / "w.npz")
    assert set(back) == set(w) and all(np.array_equal(back[k], w[k]) for k in w)

    class Evil:
        def __reduce__(self):
            return (os.system, ("touch " + str(tmp_path / "pwned"),))
    np.save(tmp_path / "evil.npy", {"pnet": [Evil()], "rnet": [], "onet": []}, allow_pickle=True)
    r = subprocess.run([sys.executable, tool, "--i-trust-this-file", str(tmp_path / "evil.npy"), str(tmp_path / "evil.npz")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "refusing to load global" in r.stderr and not (tmp_path / "pwned").exists()
