"""Host logic of the data layer mirror, the solver files and the `caffe` tool — no GPU needed."""
import os
import subprocess
import sys

import numpy as np
import pytest

from fcn_object_detector_amd import data_layer as D
from fcn_object_detector_amd import proto
from oracle import detect_ref as R

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAFFE = os.path.join(REPO, "fcn_object_detector_amd", "build", "tools", "caffe")


def test_param_str():
    p = D.parse_param_str("448,448,16,4,8,train.txt")
    assert p == dict(image_size_x=448, image_size_y=448, stride=16, num_classes=4, batch_size=8, train_fn="train.txt", mode="mask")
    assert D.parse_param_str("448, 224, 16, 1, 2, synthetic:3, detectnet")["mode"] == "detectnet"
    for bad in ("448,448,16,4,8", "a,448,16,4,8,train.txt", "1,2,3,4,5,f,other"):
        with pytest.raises(ValueError):
            D.parse_param_str(bad)


def test_dataset_textfile(tmp_path):
    # reference convention: a record on every 2nd line (data_argumentation_layer.py:166), labels re-indexed by np.unique
    fn = tmp_path / "train.txt"
    recs = [("a.jpg", "a_m.png", 7, 1, 2, 30, 40), ("b.jpg", "b_m.png", 3, 5, 6, 70, 80), ("c.jpg", "c_m.png", 7, 9, 9, 9, 9)]
    with open(fn, "w") as f:
        for r in recs:
            f.write(" ".join(str(v) for v in r) + "\n")
            f.write("# skipped line\n")
    man = tmp_path / "labels"
    imgs, masks, labels, rects = D.read_data_from_textfile2(str(fn), manifest_dir=str(man))
    assert list(imgs) == ["a.jpg", "b.jpg", "c.jpg"] and list(masks) == ["a_m.png", "b_m.png", "c_m.png"]
    assert labels.tolist() == [1, 0, 1]
    assert rects.tolist() == [[1, 2, 30, 40], [5, 6, 70, 80], [9, 9, 9, 9]]
    files = os.listdir(man)
    assert len(files) == 1 and files[0].startswith("labels_")
    assert open(man / files[0]).read() == "1 3\n2 7\n"


def test_rect_helpers_match_oracle():
    rng = np.random.default_rng(5)
    for _ in range(200):
        H, W = int(rng.integers(50, 700)), int(rng.integers(50, 700))
        rects = [[int(rng.integers(0, W - 10)), int(rng.integers(0, H - 10)), int(rng.integers(1, 200)), int(rng.integers(1, 200))]
                 for _ in range(int(rng.integers(0, 4)))]
        dst = (int(rng.integers(32, 600)), int(rng.integers(32, 600)))
        assert D.resize_rects((H, W), dst, rects) == [tuple(r) for r in R.resize_rects((H, W), dst, rects)]
        for flag in (-1, 0, 1):
            assert D.flip_rects((H, W), rects, flag) == [list(r) for r in R.flip_rects((H, W), rects, flag)]
    assert D.resize_rects((480, 640), (448, 448), [(361, 198, 100, 134)]) == [(252, 184, 69, 125)]


def test_layer_sample_geometry():
    import random
    from fcn_object_detector_amd.pylayer import TopProxy
    lay = D.DataArgumentationLayer()
    lay.param_str = "128,96,16,3,2,synthetic:3,detectnet"
    tops = [TopProxy(n) for n in ("data", "coverage-label", "bbox-label", "size-block", "obj-block", "coverage-block")]
    with pytest.raises(Exception):
        lay.setup([], tops[:3])
    lay.setup([], tops)
    lay.reshape([], tops)
    assert tops[0].shape == (2, 3, 96, 128) and tops[1].shape == (2, 3, 6, 8) and tops[5].shape == (2, 12, 6, 8)
    random.seed(3)
    lay._color_rng = np.random.default_rng(3)
    from oracle import scene_ref as S
    for zoom in (False, True):
        lay.zoom_augmentation = zoom
        for _ in range(5):
            img, mask, rects, labels = S.make_sample(lay)          # the layer's plan rendered by the oracle (no GPU here)
            assert img.shape == (96, 128, 3) and img.dtype == np.float32 and 0.0 <= img.min() and img.max() <= 1.0
            assert mask.shape == (96, 128) and set(np.unique(mask)) <= {0, 1, 2, 3}
            assert len(rects) == len(labels) >= 1
            for (x, y, w, h), lab in zip(rects, labels):
                assert w > 0 and h > 0 and 0 <= lab < 3
                if not zoom:             # (a zoomed crop may cut through its box)
                    assert 0 <= x and 0 <= y and x + w <= 128 and y + h <= 96
    lay2 = D.DataArgumentationLayer()
    lay2.param_str = "128,96,16,3,2,/nonexistent/train.txt"
    with pytest.raises(ValueError):
        lay2.setup([], tops)


def test_zoom_plan_known_answer(monkeypatch):
    """crop_image_dimension (argumentation_engine.py:190-236) by hand for a 640x480 scene, box (100,120,80,60), draws
    enlarge 2.0 / 1.5, shifts 10 / 20, signs + / -:  widths (160, 120.0), heights (120, 90.0); x = 140-160 = -20, y = 150-120 = 30,
    w = 280.0, h = 210.0; centre (150, 130) -> nx 10 -> clamped to x = -20, ny 25 then pushed down by 5 to 30; x < 0 -> 0."""
    import random
    uni, ints = iter([2.0, 1.5]), iter([10, 20, 1, 0])
    asked = []
    monkeypatch.setattr(random, "uniform", lambda a, b: (asked.append(("u", a, b)), next(uni))[1])
    monkeypatch.setattr(random, "randint", lambda a, b: (asked.append(("i", a, b)), next(ints))[1])
    window, rect = D.plan_zoom((480, 640), (100, 120, 80, 60), (100, 120, 80, 60))
    assert window == (0, 30, 280, 210) and rect == [100, 90, 80, 60]
    assert asked == [("u", 1.0, 8.0), ("u", 1.0, 8.0), ("i", 0, 140), ("i", 0, 105), ("i", 0, 1), ("i", 0, 1)]
    assert D.plan_zoom((480, 640), (0, 0, 0, 10), (0, 0, 0, 10)) is None          # the reference divides by the box width


def test_zoom_plans_stay_inside_the_scene():
    import random
    random.seed(12)
    rng = np.random.default_rng(12)
    n = 0
    for _ in range(500):
        w, h = int(rng.integers(8, 400)), int(rng.integers(8, 300))
        r = (int(rng.integers(0, 640 - w)), int(rng.integers(0, 480 - h)), w, h)
        z = D.plan_zoom((480, 640), r, r)
        if z is None:
            continue
        n += 1
        (x, y, vw, vh), nr = z
        assert 0 <= x and 0 <= y and vw > 0 and vh > 0 and x + vw <= 640 and y + vh <= 480
        assert nr[2:] == [w, h] and nr[0] <= r[0] and nr[1] <= r[1]
    assert n > 400


def test_color_plan_ranges():
    rng = np.random.default_rng(0)
    kinds, per_channel = set(), 0
    for _ in range(300):
        c = D.plan_color(rng)
        b = c["blur"]
        kinds.add((b["kind"], b.get("k")))
        assert b["kind"] != "gauss" or 0.0 <= b["sigma"] <= 3.0
        assert 0 <= c["sharpen"][0] <= 1 and 0.75 <= c["sharpen"][1] <= 1.5 and 0 <= c["gray"] <= 0.5
        assert all(-2 <= a <= 21 for a in c["add"]) and all(0.75 <= m <= 1.25 for m in c["mul"])
        per_channel += len(set(c["add"])) > 1
    assert {("median", 3), ("median", 5), ("median", 7), ("box", 2), ("box", 7), ("gauss", None)} <= kinds
    assert ("median", 4) not in kinds and 90 < per_channel < 210
    t = D.gauss_taps(3.0)
    assert len(t) == 11 and abs(float(t[0] + 2 * t[1:].sum()) - 1.0) < 1e-6 and len(D.gauss_taps(0.2)) == 2


def test_oracle_blurs_against_scipy():
    """The oracle's restated imgaug operators against scipy.ndimage's filters of the same definition (independent code)."""
    from scipy import ndimage
    from oracle import scene_ref as S
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (37, 29, 3), dtype=np.uint8)
    for sigma in (0.4, 1.3, 3.0):
        taps = D.gauss_taps(sigma)
        want = np.stack([ndimage.gaussian_filter(img[..., c].astype(np.float64), sigma, mode="mirror", truncate=(len(taps) - 1) / sigma)
                         for c in range(3)], -1)
        got = S.blur_gauss(img, taps).astype(np.float64)
        assert np.abs(got - want).max() <= 0.5 + 5e-3          # float32 taps vs float64
    for k in (3, 5, 7):
        want = np.stack([ndimage.median_filter(img[..., c], size=k, mode="nearest") for c in range(3)], -1)
        assert np.array_equal(S.blur_median(img, k), want)
        box = np.stack([ndimage.uniform_filter(img[..., c].astype(np.float64), size=k, mode="mirror") for c in range(3)], -1)
        assert np.abs(S.blur_box(img, k).astype(np.float64) - box).max() <= 0.5 + 1e-9
    # even box kernels: the anchor is k/2, i.e. the window reaches one pixel further up/left
    flat = np.zeros((6, 6, 3), np.uint8)
    flat[2, 2] = 200
    b2 = S.blur_box(flat, 2)
    assert b2[2, 2, 0] == 50 and b2[3, 3, 0] == 50 and b2[1, 1, 0] == 0
    # identity settings: no sharpening, no offset, unit gain, no grey
    same = S.color_point_ops(img, (0.0, 1.0), [0, 0, 0], [1.0, 1.0, 1.0], 0.0)
    assert np.array_equal(same, img)
    grey = S.color_point_ops(img, (0.0, 1.0), [0, 0, 0], [1.0, 1.0, 1.0], 0.5)
    g = (img[..., 0].astype(int) * 4899 + img[..., 1].astype(int) * 9617 + img[..., 2].astype(int) * 1868 + 8192) >> 14
    assert np.abs(grey.astype(float) - (0.5 * img + 0.5 * g[..., None])).max() <= 0.5


def test_solverstate_roundtrip():
    hist = [np.arange(6, dtype=np.float32).reshape(2, 3), np.ones(4, np.float32)]
    buf = proto.pack_solverstate(17, hist, learned_net="snap_iter_17.caffemodel")
    it, h2, learned = proto.unpack_solverstate(buf, with_learned_net=True)
    assert it == 17 and learned == "snap_iter_17.caffemodel"
    assert all(np.array_equal(a, b) for a, b in zip(hist, h2))
    assert proto.unpack_solverstate(buf)[0] == 17


def test_caffe_tool_argument_errors(tmp_path):
    def run(*args):
        return subprocess.run([sys.executable, CAFFE] + list(args), capture_output=True, text=True, timeout=120)
    r = run()
    assert r.returncode == 1 and "caffe train --solver" in r.stderr
    r = run("train")
    assert r.returncode == 1 and "Need a solver definition to train" in r.stderr
    r = run("train", "--solver=s", "--weights=w", "--snapshot=x")
    assert r.returncode == 1 and "not both" in r.stderr
    r = run("time")
    assert r.returncode == 1 and "Need a model definition" in r.stderr


def test_reference_module_name_resolves():
    sys.path.insert(0, os.path.join(REPO, "fcn_object_detector_amd", "python"))
    try:
        import data_argumentation_layer as m
        assert m.DataArgumentationLayer is D.DataArgumentationLayer
    finally:
        sys.path.pop(0)


def test_label_manifest_round_trip(tmp_path):
    from fcn_object_detector_amd.detector import load_label_manifest
    fn = tmp_path / "train.txt"
    with open(fn, "w") as f:
        for lab in (12, 5, 12, 9):
            f.write("a.jpg m.png %d 1 2 3 4\n\n" % lab)
    D.read_data_from_textfile2(str(fn), manifest_dir=str(tmp_path / "labels"))
    manifest = os.path.join(str(tmp_path / "labels"), os.listdir(tmp_path / "labels")[0])
    assert load_label_manifest(manifest) == ["5", "9", "12"]             # the data layer's 2-field lines
    three = tmp_path / "three.txt"
    three.write_text("1 0 mug\n2 0 bottle\n")
    assert load_label_manifest(str(three)) == ["mug", "bottle"]          # the node's 3-field lines
    assert load_label_manifest(None, 3) == ["object_-1", "object_0", "object_1"]
