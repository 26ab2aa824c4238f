"""Pins the CPU oracle: torch-CPU cross-checks of the Caffe layer restatement, hand-checkable
groupRectangles cases, and the survey's known-answer tests for target generation (CPU only)."""
import math

import numpy as np
import pytest

from oracle import caffe_ref as R
from oracle import detect_ref as D

torch = pytest.importorskip("torch")
F = torch.nn.functional


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.mark.parametrize("cin,cout,k,s,p,h,w", [(3, 8, 7, 2, 3, 23, 31), (16, 12, 3, 1, 1, 14, 9), (8, 5, 5, 1, 2, 11, 11),
                                               (12, 7, 1, 1, 0, 6, 6)])
def test_conv_vs_torch(cin, cout, k, s, p, h, w):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, cin, h, w)).astype(np.float32)
    wt = rng.standard_normal((cout, cin, k, k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    y = R.conv2d(x, wt, b, p, s)
    ref = F.conv2d(t(x), t(wt), t(b), stride=s, padding=p).numpy()
    assert y.shape == ref.shape and np.allclose(y, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("k,s,p,h,w", [(3, 2, 0, 224 // 8, 28), (3, 2, 0, 15, 14), (3, 1, 1, 9, 7), (2, 2, 0, 8, 6), (3, 2, 1, 10, 11)])
def test_maxpool_vs_torch(k, s, p, h, w):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 5, h, w)).astype(np.float32)
    y, idx = R.max_pool(x, k, s, p, return_index=True)
    ref, ridx = F.max_pool2d(t(x), k, s, p, ceil_mode=True, return_indices=True)
    assert y.shape == tuple(ref.shape) == (2, 5, R.pool_out(h, k, p, s), R.pool_out(w, k, p, s))
    assert np.array_equal(y, ref.numpy()) and np.array_equal(idx, ridx.numpy())


def test_maxpool_first_max_wins():
    x = np.zeros((1, 1, 3, 3), np.float32)
    x[0, 0, 0, 2] = x[0, 0, 1, 0] = 5.0
    y, idx = R.max_pool(x, 3, 1, 0, return_index=True)
    assert y[0, 0, 0, 0] == 5.0 and idx[0, 0, 0, 0] == 2          # raster-first maximum


def test_avepool_vs_torch():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((1, 4, 56, 56)).astype(np.float32)
    for k in (56, 28, 14, 8):
        y = R.ave_pool(x, k, k, 0)
        ref = F.avg_pool2d(t(x), k, k, 0, ceil_mode=True).numpy()
        assert np.allclose(y, ref, rtol=1e-5, atol=1e-6)
    # padded case: Caffe divides by the window clipped to H+pad (padding counted)
    y = R.ave_pool(x[:, :, :6, :6], 3, 2, 1)
    ref = F.avg_pool2d(t(x[:, :, :6, :6]), 3, 2, 1, ceil_mode=True, count_include_pad=True).numpy()
    assert y.shape == ref.shape
    assert np.allclose(y[:, :, :-1, :-1], ref[:, :, :-1, :-1], rtol=1e-5, atol=1e-6)


def test_lrn_vs_torch():
    rng = np.random.default_rng(4)
    x = (rng.standard_normal((2, 64, 6, 5)) * 30).astype(np.float32)
    y = R.lrn_across(x, 5, 1e-4, 0.75, 1.0)
    ref = F.local_response_norm(t(x), 5, alpha=1e-4, beta=0.75, k=1.0).numpy()
    assert np.allclose(y, ref, rtol=1e-5, atol=1e-6)


def test_deconv_vs_torch_and_bilinear():
    rng = np.random.default_rng(5)
    for c, k, s, p, h in [(4, 8, 4, 2, 5), (3, 4, 2, 1, 7), (2, 16, 8, 4, 3)]:
        x = rng.standard_normal((2, c, h, h)).astype(np.float32)
        w = R.bilinear_filler((c, 1, k, k))
        y = R.deconv2d(x, w, None, p, s, group=c)
        ref = F.conv_transpose2d(t(x), t(w), None, stride=s, padding=p, groups=c).numpy()
        assert y.shape == ref.shape == (2, c, s * (h - 1) + k - 2 * p, s * (h - 1) + k - 2 * p)
        assert np.allclose(y, ref, rtol=1e-5, atol=1e-5)
    assert np.allclose(R.bilinear_filler((1, 1, 4, 4))[0, 0, 0], [0.0625, 0.1875, 0.1875, 0.0625])


def test_sigmoid_softmax_losses():
    rng = np.random.default_rng(6)
    x = rng.standard_normal((3, 4, 5, 5)).astype(np.float32)
    assert np.allclose(R.sigmoid(x), torch.sigmoid(t(x)).numpy(), atol=1e-6)
    assert np.allclose(R.softmax(x), F.softmax(t(x), 1).numpy(), atol=1e-6)
    a, b = x, rng.standard_normal(x.shape).astype(np.float32)
    assert math.isclose(R.l1_loss(a, b), float(np.abs(a - b).sum() / 3), rel_tol=1e-6)
    assert math.isclose(R.euclidean_loss(a, b), float(((a - b) ** 2).sum() / 6), rel_tol=1e-6)
    lab = rng.integers(0, 4, (3, 1, 5, 5))
    ce = F.cross_entropy(t(x), t(lab[:, 0]), reduction="sum").item()
    assert math.isclose(R.softmax_loss(x, lab, normalize=False), ce / 3, rel_tol=1e-5)
    assert math.isclose(R.softmax_loss(x, lab, normalize=True), ce / 75, rel_tol=1e-5)


# ------------------------------------------------------------------ groupRectangles (hand-checkable)

def test_cv_round_half_even():
    assert [D.cv_round(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]
    assert D.to_rect([10.5, 11.5, -0.5, 3.7]) == (10, 12, 0, 4)
    assert D.to_rect([10.5, 11.5, -0.5, 3.7], "trunc") == (10, 11, 0, 3)


def test_group_two_clusters_and_small_one_dropped():
    a = [(100, 100, 50, 60), (101, 100, 50, 60), (100, 101, 51, 60), (99, 100, 50, 61)]      # n = 4 > 3 -> kept
    b = [(300, 40, 80, 80), (301, 41, 80, 80), (300, 40, 81, 79), (299, 40, 80, 80), (300, 39, 80, 81)]
    c = [(10, 10, 20, 20), (10, 11, 20, 20), (11, 10, 20, 20)]                                # n = 3 <= 3 -> dropped
    rects, weights = D.group_rectangles(a + c + b, 3, 0.2)
    assert weights == [4, 5]
    assert rects[0] == (100, 100, 50, 60)        # sums 400 401 201 241 -> *0.25f -> 100 100.25 50.25 60.25
    assert rects[1] == (300, 40, 80, 80)         # sums 1500 200 401 400 -> *0.2f  -> 300 40 80.2 80
    # classes are numbered by first member: interleave and the order follows the first occurrence
    rects2, weights2 = D.group_rectangles([b[0], a[0]] + b[1:] + a[1:], 3, 0.2)
    assert weights2 == [5, 4]


def test_group_mean_rounds_half_to_even():
    # sums: x 8*4+2 = 34? use 4 rects with x = 10,10,11,11 -> sum 42 * 0.25f = 10.5 -> 10 (even); y sum 46 -> 11.5 -> 12
    rs = [(10, 11, 40, 40), (10, 11, 40, 40), (11, 12, 40, 40), (11, 12, 40, 40)]
    rects, weights = D.group_rectangles(rs, 3, 0.2)
    assert weights == [4] and rects[0] == (10, 12, 40, 40)


def test_group_nested_small_cluster_removed():
    big = [(100, 100, 200, 200)] * 6 + [(101, 101, 200, 200)] * 2          # n = 8
    small = [(150, 150, 50, 50)] * 4                                           # n = 4, inside big, n2 > max(3, n1)
    rects, weights = D.group_rectangles(big + small, 3, 0.2)
    assert weights == [8] and rects[0] == (100, 100, 200, 200)
    # a nested cluster with MORE members than the outer one survives
    rects, weights = D.group_rectangles(big[:4] + [(150, 150, 50, 50)] * 6, 3, 0.2)
    assert sorted(weights) == [4, 6]


def test_group_threshold_zero_passthrough_and_chain():
    rs = [(0, 0, 10, 10), (50, 50, 10, 10)]
    assert D.group_rectangles(rs, 0, 0.2) == (rs, [1, 1])
    # similarity is not transitive: a chain 0-1-2-3-4 is ONE class although ends are not similar
    chain = [(100 + 6 * i, 100, 60, 60) for i in range(5)]       # delta = 0.2*(60+60)/2 = 12 -> neighbours (6) and next (12) similar
    assert not D.similar_rects(chain[0], chain[4], 0.2)
    labels, n = D.partition(chain, 0.2)
    assert n == 1 and labels == [0] * 5


def test_partition_fast_matches_literal():
    rng = np.random.default_rng(7)
    for trial in range(20):
        n = int(rng.integers(1, 120))
        base = rng.integers(0, 300, (max(n // 6, 1), 2))
        pick = rng.integers(0, len(base), n)
        rects = np.concatenate([base[pick] + rng.integers(-6, 7, (n, 2)), rng.integers(30, 60, (n, 2))], axis=1)
        lab, nc = D.partition([tuple(int(v) for v in r) for r in rects], 0.2)
        lab2, nc2 = D.partition_fast(rects, 0.2)
        assert nc == nc2 and list(lab2) == lab
        r1 = D.group_rectangles([tuple(int(v) for v in r) for r in rects], 3, 0.2)
        r2 = D.group_rectangles([tuple(int(v) for v in r) for r in rects], 3, 0.2, fast=True)
        assert r1 == r2


def test_vote_boxes_quirks():
    # (x1,y1,x2,y2) is handed over as (x,y,w,h); height filter uses rect[3]-rect[1]
    boxes = np.array([[100.4, 100.6, 160.5, 130.2]] * 4)
    det = D.vote_boxes(boxes, 3, 0.2)
    assert det == [[100, 101, 160, 130, math.log(4)]]                 # 160.5 -> 160 (half to even)
    assert D.vote_boxes(boxes, 3, 0.2, round_mode="trunc") == [[100, 100, 160, 130, math.log(4)]]
    assert D.vote_boxes(np.array([[100.0, 100.0, 160.0, 119.0]] * 4), 3, 0.2) == []      # 119-100 < 20
    assert D.vote_boxes(np.zeros((4, 4)), 3, 0.2) == [] and D.vote_boxes(np.zeros((0, 4)), 3, 0.2) == []


def test_gridbox_to_boxes_order_and_dtype():
    cvg = np.zeros((28, 28), np.float32)
    bb = np.zeros((4, 28, 28), np.float32)
    cvg[3, 5], cvg[2, 9] = 0.5, 0.9
    bb[:, 3, 5] = [-1.5, -2.25, 10.5, 20.0]
    bb[:, 2, 9] = [1, 2, 3, 4]
    boxes, mask = D.gridbox_to_boxes(cvg, bb, 0.5, 448, 448, 16)
    assert boxes.dtype == np.float64 and mask.sum() == 2
    assert boxes.tolist() == [[9 * 16 + 1, 2 * 16 + 2, 9 * 16 + 3, 2 * 16 + 4], [5 * 16 - 1.5, 3 * 16 - 2.25, 5 * 16 + 10.5, 3 * 16 + 20.0]]


# ------------------------------------------------------------------ target generation KATs

def test_kat1_single_rect():
    fg, bl, sl, ol, cl = D.bounding_box_parameterized_labels(448, 448, [(100, 120, 80, 60)], [0], 16, 1)
    pos = np.argwhere(fg[0] == 1)
    assert len(pos) == 28 and set(pos[:, 0]) == set(range(7, 12)) and set(pos[:, 1]) == set(range(6, 12))
    # SURVEY's cell "(i=7, j=8)" = column 7, row 8
    assert bl[:, 8, 7].tolist() == [-12.0, -8.0, 68.0, 52.0]
    assert np.allclose(sl[:, 8, 7], [0.0125, 1 / 60.0, 0.0125, 1 / 60.0])
    assert np.allclose(ol[:, 8, 7], 0.05333333)
    cell = lambda i, j: (i * 16.0, j * 16.0, 16.0, 16.0)
    assert abs(D.jaccard_iou(cell(6, 7), (100, 120, 80, 60)) - 0.315126) < 1e-6
    assert abs(D.jaccard_iou(cell(11, 7), (100, 120, 80, 60)) - 0.095908) < 1e-6     # below the 0.1 threshold
    assert abs(D.jaccard_iou(cell(7, 11), (100, 120, 80, 60)) - 0.208333) < 1e-6


def test_kat2_two_rects_eleven_classes():
    fg, bl, sl, ol, cl = D.bounding_box_parameterized_labels(448, 448, [(40, 64, 120, 200), (300, 310, 64, 48)], [3, 10], 8, 11)
    assert [int(fg[c].sum()) for c in range(11)] == [0, 0, 0, 375, 0, 0, 0, 0, 0, 0, 63]
    assert bl.sum() == 7260.0 and abs(sl.sum() - 14.593750) < 1e-6 and abs(ol.sum() - 9.25) < 1e-6 and cl.sum() == 1752


def test_rect_helpers():
    # f32 arithmetic: xb - xt = 322.69998 - 252.7 = 69.99998 -> int() truncates to 69 (exact math would give 70)
    assert D.resize_rects((480, 640), (448, 448), [(361, 198, 100, 134)]) == [(252, 184, 69, 125)]
    assert D.flip_rects((480, 640), [[10, 20, 30, 40]], 1) == [[599, 20, 30, 40]]
    assert D.flip_rects((480, 640), [[10, 20, 30, 40]], 0) == [[10, 419, 30, 40]]
    assert D.flip_rects((480, 640), [[10, 20, 30, 40]], -1) == [[599, 419, 30, 40]]


def test_preprocess_identity_resize():
    rng = np.random.default_rng(0)
    frame = rng.integers(0, 256, (32, 48, 3), dtype=np.uint8)
    blob = D.preprocess_frame(frame, 48, 32)
    im = D.demean_rgb_image(frame)
    assert blob.shape == (3, 32, 48) and blob.min() == 0.0 and blob.max() == 1.0
    assert np.array_equal(blob, im.transpose(2, 0, 1).astype(np.float32))
    up = D.preprocess_frame(frame, 96, 64)
    ref = F.interpolate(t(im.transpose(2, 0, 1)[None]), size=(64, 96), mode="bilinear", align_corners=False)[0].numpy()
    assert np.allclose(up, ref, atol=1e-6)


def test_scene_image_helpers():
    from oracle import scene_ref as S
    rng = np.random.default_rng(0)
    im = rng.integers(0, 256, (7, 9, 3), dtype=np.uint8)
    assert np.array_equal(S.flip_image(im, 0), im[::-1]) and np.array_equal(S.flip_image(im, 1), im[:, ::-1])
    assert np.array_equal(S.flip_image(im, -1), im[::-1, ::-1])
    d = S.demean_rgb_image(im)
    assert d.dtype == np.float32 and d.min() == 0.0 and d.max() == 1.0
    ref = im.astype(np.float32) - np.array(S.MEAN_BGR, np.float32)
    assert np.allclose(d, (ref - ref.min()) / (ref.max() - ref.min()), atol=1e-6)
    assert np.array_equal(S.resize_bilinear(im, 9, 7), im)
    flat = np.full((10, 12, 3), 77, np.uint8)
    assert np.all(S.resize_bilinear(flat, 30, 5) == 77)
    # 2x upscale of a ramp: OpenCV half-pixel centres
    ramp = np.array([[0.0, 10.0]], np.float32)
    assert np.allclose(S.resize_bilinear(ramp, 4, 1), [[0.0, 2.5, 7.5, 10.0]])
    m = np.arange(12, dtype=np.uint8).reshape(3, 4)
    assert np.array_equal(S.resize_nearest(m, 8, 6), m.repeat(2, 0).repeat(2, 1))




# ---- oracle/caffe_cpu.c (compiled loops of the memory-bound layers) against the numpy statement of the same layers ----

def _numpy_twin(monkeypatch):
    monkeypatch.setattr(R, "_C", None)


@pytest.mark.skipif(R._C is None, reason="oracle/libcaffe_cpu.so not built (make -C oracle)")
@pytest.mark.parametrize("c,h,w,k,s,p", [(3, 23, 31, 7, 2, 3), (16, 14, 9, 3, 1, 1), (8, 11, 11, 5, 1, 2), (4, 6, 8, 2, 2, 0)])
def test_c_im2col_equals_numpy(monkeypatch, c, h, w, k, s, p):
    x = np.random.default_rng(3).standard_normal((c, h, w)).astype(np.float32)
    got = R.im2col(x, k, k, p, p, s, s)
    _numpy_twin(monkeypatch)
    assert np.array_equal(got, R.im2col(x, k, k, p, p, s, s))


@pytest.mark.skipif(R._C is None, reason="oracle/libcaffe_cpu.so not built (make -C oracle)")
@pytest.mark.parametrize("k,s,p,h,w", [(3, 2, 0, 28, 28), (3, 2, 0, 15, 14), (3, 1, 1, 9, 7), (2, 2, 0, 8, 6), (3, 2, 1, 10, 11), (3, 1, 1, 1, 1)])
def test_c_maxpool_equals_numpy(monkeypatch, k, s, p, h, w):
    x = np.random.default_rng(4).standard_normal((2, 5, h, w)).astype(np.float32)
    x[0, 0, : h // 2] = 0.25      # ties: the first maximum in raster order wins
    y0 = R.max_pool(x, k, s, p)
    y1, i1 = R.max_pool(x, k, s, p, return_index=True)
    _numpy_twin(monkeypatch)
    r0 = R.max_pool(x, k, s, p)
    r1, j1 = R.max_pool(x, k, s, p, return_index=True)
    assert np.array_equal(y0, r0) and np.array_equal(y1, r1) and np.array_equal(i1, j1) and np.array_equal(y0, y1)


@pytest.mark.skipif(R._C is None, reason="oracle/libcaffe_cpu.so not built (make -C oracle)")
@pytest.mark.parametrize("c,n,alpha,beta", [(7, 5, 1e-4, 0.75), (16, 3, 2e-2, 0.5), (3, 5, 1.0, 0.75)])
def test_c_lrn_equals_numpy(monkeypatch, c, n, alpha, beta):
    x = (np.random.default_rng(5).standard_normal((2, c, 6, 5)) * 3).astype(np.float32)
    y, sc = R.lrn_across(x, n, alpha, beta, 1.0, return_scale=True)
    y_only = R.lrn_across(x, n, alpha, beta, 1.0)
    _numpy_twin(monkeypatch)
    ry, rsc = R.lrn_across(x, n, alpha, beta, 1.0, return_scale=True)
    assert np.array_equal(sc, rsc) and np.array_equal(y, ry)      # same additions in the same order, no fused multiply-add; numpy's power
    assert np.allclose(y_only, ry, rtol=3e-7, atol=0)             # TEST phase: powf against numpy's float32 power, a last-bit difference at most


def test_detection_window_geometry_by_hand():
    """detection_window_roi (fcn_object_detector.py:257-277): oracle restatement == the product's host mirror == the rectangles
    worked out by hand for a 480 x 640 frame, and for odd sizes (Python-2 integer division)."""
    from fcn_object_detector_amd.detector import detection_window_roi
    from oracle import detect_ref as D
    frame = np.random.default_rng(1).integers(0, 256, (480, 640, 3), dtype=np.uint8)
    rois, rects = D.detection_window_roi(D.demean_rgb_image(frame), (32, 24), 2)
    assert [tuple(int(v) for v in r) for r in rects] == [(0, 0, 320, 240), (320, 0, 320, 240), (0, 240, 320, 240), (320, 240, 320, 240),
                                                         (160, 120, 320, 240)]
    assert all(r.shape == (3, 24, 32) for r in rois)
    assert np.array_equal(detection_window_roi(frame.shape, 2), np.asarray(rects, np.int32))
    # the node calls it with stride 1: the whole frame twice (the central crop of a full-size window starts at 0)
    assert detection_window_roi((301, 517, 3), 1).tolist() == [[0, 0, 517, 301], [0, 0, 517, 301]]
    assert detection_window_roi((301, 517, 3), 2).tolist()[-1] == [517 // 2 - 258 // 2, 301 // 2 - 150 // 2, 258, 150]
    with pytest.raises(ValueError):
        detection_window_roi((1, 5, 3), 2)
    # a window's boxes in frame coordinates: scaled by window / net size (truncating), moved by the window's origin
    b = D.window_boxes_to_frame((160, 120, 320, 240), np.array([[10, 20, 100, 80, 2]]), 192, 160)
    assert b.tolist() == [[160 + int(10 * 320 / 192), 120 + int(20 * 240 / 160), 160 + int(100 * 320 / 192), 120 + int(80 * 240 / 160), 2]]
