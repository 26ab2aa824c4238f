"""The oracle (and the product's host helpers) held to vectors the REFERENCE ITSELF computed.

tests/golden/reference_numpy_rows.npz was written by tests/golden/make_reference_golden.py, which executes the reference's
own function bodies (read from /root/reference at run time, build container only) for the pure-numpy rows of SURVEY.md
section 8a: A4 (JaccardCoeff.iou, bounding_box_parameterized_labels, generate_box_labels, grid_region -
argumentation_engine.py:24-109,272-292), A5 (rect arithmetic of resize_image_and_labels / flip_image, demean_rgb_image -
:114-138,241-267,297-303), A6 (the node's float64 demean, fcn_object_detector.py:407-413), A7 (gridbox_to_boxes,
:357-394, and its stride-16 twin boundary_refinement.py:265-302) and A9 (resize_detection, :396-405).  These rows are
therefore PINNED; A1 / A2 / A8 (Caffe, OpenCV) are not.  CPU only; the -m gpu twins live in
tests/test_golden.py.
"""
import os

import numpy as np
import pytest

from fcn_object_detector_amd import data_layer, detector
from oracle import detect_ref as D

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "golden", "reference_numpy_rows.npz")


@pytest.fixture(scope="module")
def G():
    return np.load(PATH)


def a4_cases(G):
    for name in G["a4_names"]:
        name = str(name)
        h, w, s, c = (int(v) for v in G["a4_%s_meta" % name])
        yield name, h, w, s, c, [tuple(int(v) for v in r) for r in G["a4_%s_rects" % name]], [int(v) for v in G["a4_%s_labels" % name]]


def test_iou_scores(G):
    with np.errstate(all="ignore"):
        got = np.array([float(D.jaccard_iou(c, tuple(int(v) for v in r))) for c, r in zip(G["iou_cells"], G["iou_rects"])])
    assert np.array_equal(got, G["iou_scores"])
    assert (G["iou_scores"] > 0.1).sum() > 50 and (G["iou_scores"] == 0).sum() > 50      # both outcomes are exercised


def test_target_generation(G):
    n = 0
    for name, h, w, s, c, rects, labels in a4_cases(G):
        with np.errstate(all="ignore"):
            got = D.bounding_box_parameterized_labels(h, w, rects, labels, s, c)
        for key, arr in zip(("fg", "bbox", "size", "obj", "cvg"), got):
            ref = G["a4_%s_%s" % (name, key)]
            assert arr.shape == ref.shape and np.array_equal(arr, ref, equal_nan=True), (name, key)
        assert np.array_equal(D.grid_region(h, w, s), G["a4_%s_grid" % name]), name
        n += 1
    assert n >= 10
    # the survey's known answers hold on the reference's own output
    assert int(G["a4_kat1_cvg"][0].sum()) == 28 and np.array_equal(G["a4_kat1_bbox"][:, 8, 7], [-12, -8, 68, 52])      # cell i = 7 (x), j = 8 (y)
    assert [int(G["a4_kat2_fg"][k].sum()) for k in range(11)] == [0, 0, 0, 375, 0, 0, 0, 0, 0, 0, 63]
    assert G["a4_kat2_bbox"].sum() == 7260.0 and G["a4_kat2_cvg"].sum() == 1752
    # a zero-area rect scores 0 / inf = 0 in the reference and marks no cell (so its unguarded 1 / w is never evaluated)
    assert np.isfinite(G["a4_zero_area_size"]).all() and set(np.unique(G["a4_zero_area_size"])) == {0.0, 1.0 / 12, 1.0 / 9}


def _split(flat, counts):
    out, p = [], 0
    for n in counts:
        out.append([tuple(int(v) for v in r) for r in flat[p:p + n]])
        p += n
    return out


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_resize_and_flip_rects(G, impl):
    resize = D.resize_rects if impl == "oracle" else data_layer.resize_rects
    flip = D.flip_rects if impl == "oracle" else data_layer.flip_rects
    ins, outs = _split(G["a5_resize_in"], G["a5_resize_counts"]), _split(G["a5_resize_out"], G["a5_resize_counts"])
    for hw, net, rin, rout in zip(G["a5_resize_src_hw"], G["a5_resize_net_wh"], ins, outs):
        got = resize((int(hw[0]), int(hw[1])), (int(net[0]), int(net[1])), rin)
        assert [tuple(int(v) for v in r) for r in got] == rout
    ins, outs = _split(G["a5_flip_in"], G["a5_flip_counts"]), _split(G["a5_flip_out"], G["a5_flip_counts"])
    for hw, flag, rin, rout in zip(G["a5_flip_hw"], G["a5_flip_flags"], ins, outs):
        got = flip((int(hw[0]), int(hw[1])), rin, int(flag))
        assert [tuple(int(v) for v in r) for r in got] == rout
    assert (G["a5_flip_out"][:, :2] == 0).any()      # the clamp-to-zero branch is in the fixture


def test_demean(G):
    for i in range(3):
        im = G["demean_%d_in" % i]
        assert np.array_equal(D.demean_rgb_image(im, np.float32), G["a5_demean_%d_f32" % i])
        assert np.array_equal(D.demean_rgb_image(im, np.float64), G["a6_demean_%d_f64" % i])


def a7_cases(G):
    i = 0
    while "a7_%d_s8_meta" % i in G:
        for tag in ("s8", "s16"):
            key = "a7_%d_%s" % (i, tag)
            net_w, net_h, stride = (int(v) for v in G[key + "_meta"])
            yield key, net_w, net_h, stride, float(G[key + "_thresh"]), G[key + "_cvg"], G[key + "_bbox"]
        i += 1


def test_gridbox_to_boxes(G):
    n = fired = 0
    for key, net_w, net_h, stride, thresh, cvg, bb in a7_cases(G):
        boxes, mask = D.gridbox_to_boxes(cvg, bb, thresh, net_w, net_h, stride)
        assert np.array_equal(np.asarray(boxes, np.float64).reshape(-1, 4), G[key + "_boxes"]), key
        assert np.array_equal(np.asarray(mask, bool), G[key + "_mask"]), key
        yx = np.argwhere(G[key + "_mask"])      # the reference's (x, y, p) side output (unused downstream) is in np.where order too
        assert np.array_equal(G[key + "_cvgs"][:, :2], yx[:, ::-1]) and np.array_equal(G[key + "_cvgs"][:, 2], cvg[yx[:, 0], yx[:, 1]].astype(np.float64))
        n += 1
        fired += len(G[key + "_boxes"])
    assert n == 10 and fired > 4000


def test_resize_detection(G):
    i = 0
    while "a9_%d_in" % i in G:
        in_size = tuple(int(v) for v in G["a9_%d_in_size" % i])
        assert np.array_equal(D.resize_detection(in_size, G["a9_%d_in" % i].copy(), 448, 448), G["a9_%d_out" % i])
        assert np.array_equal(detector.resize_detection(in_size, G["a9_%d_in" % i].copy(), 448, 448), G["a9_%d_out" % i])
        i += 1
    assert i == 4
