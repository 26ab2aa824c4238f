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
    from oracle import scene_ref as S
    for _ in range(5):
        img, mask, rects, labels = S.make_sample(lay)          # the layer's plan rendered by the oracle (no GPU here)
        assert img.shape == (96, 128, 3) and img.dtype == np.float32 and 0.0 <= img.min() and img.max() <= 1.0
        assert mask.shape == (96, 128) and set(np.unique(mask)) <= {0, 1, 2, 3}
        assert len(rects) == len(labels) >= 1
        for (x, y, w, h), lab in zip(rects, labels):
            assert 0 <= x and 0 <= y and x + w <= 128 and y + h <= 96 and w > 0 and h > 0 and 0 <= lab < 3
    lay2 = D.DataArgumentationLayer()
    lay2.param_str = "128,96,16,3,2,/nonexistent/train.txt"
    with pytest.raises(ValueError):
        lay2.setup([], tops)


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
