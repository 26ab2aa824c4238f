"""run_detector2 after net.forward() on the device (csrc/mask.hip) against oracle/mask_ref.py, bit for bit (-m gpu):
the frame-sized probability map and the padded bounding rectangle of the largest contour per (window, class)
(reference: scripts/fcn_object_detector.py:208-236, create_mask_labels :279-303)."""
import numpy as np
import pytest

from fcn_object_detector_amd import models, proto
from fcn_object_detector_amd.detector import FCNObjectDetector, HeadMapping, detection_window_roi, score_masks_from_maps
from fcn_object_detector_amd.engine import Engine
from fcn_object_detector_amd.netspec import NetSpec, fill_params
from oracle import mask_ref as M

pytestmark = pytest.mark.gpu


def same(got, want):
    pg, bg = got
    pw, bw = want
    assert pg.shape == pw.shape and np.array_equal(pg, pw)
    assert len(bg) == len(bw), (len(bg), len(bw))
    for (rg, cg), (rw, cw) in zip(bg, bw):
        assert cg == cw and np.array_equal(np.asarray(rg), np.asarray(rw)), (rg, cg, rw, cw)
    return len(bw)


def masks_to_scores(rows_per_class, n=1):
    """hand-drawn masks as score maps (1.0 where '#'), class 0 = background"""
    h, w = len(rows_per_class[0]), len(rows_per_class[0][0])
    fm = np.zeros((n, 1 + len(rows_per_class), h, w), np.float32)
    for c, rows in enumerate(rows_per_class):
        fm[:, c + 1] = np.array([[1.0 if ch == "#" else 0.0 for ch in r] for r in rows], np.float32)
    return fm


def test_hand_checkable_masks_identity_resize(gpu):
    """Masks drawn by hand, window size = map size (the resize is the identity): blocks, a lone pixel, a line, a diagonal chain
    (8-connectivity), a ring (its hole border ties with the outer one), equal areas (the later component wins), a nested component."""
    cases = [
        [".....", "..##.", "..##.", "....."],
        [".....", "..#..", ".....", "....."],
        [".....", ".###.", ".....", "....."],
        ["#....", ".#...", "..#..", "...##"],
    ]
    fm = masks_to_scores(cases)
    got = score_masks_from_maps(fm, [(3, 2, 5, 4)], (9, 11), 0.5)
    want = M.run_detector2_post(fm, [(3, 2, 5, 4)], (9, 11), 0.5)
    assert same(got, want) == 1      # the lone pixel, the line and the diagonal chain (a path walked out and back) enclose no area
    assert got[1][0][1] == 1 and got[1][0][0].tolist() == [2 + 3 - 10, 1 + 2 - 10, 2 + 20, 2 + 20]
    big = [
        [".......", ".#####.", ".#...#.", ".#...#.", ".#####.", "......."],
        ["##.....", "##.....", ".......", "...##..", "...##..", "......."],
        ["###....", "###....", "###....", "......#", ".....##", "......."],
        ["#######", "#.....#", "#.###.#", "#.###.#", "#.....#", "#######"],
    ]
    fm = masks_to_scores(big, n=2)
    rects = [(0, 0, 7, 6), (5, 3, 7, 6)]
    got = score_masks_from_maps(fm, rects, (10, 13), 0.5)
    assert same(got, M.run_detector2_post(fm, rects, (10, 13), 0.5)) == 8
    assert got[1][1][0].tolist() == [3 - 10, 3 - 10, 22, 22]              # equal areas: the component found last


@pytest.mark.parametrize("shape,win,frame,stride", [((5, 5, 28, 28), (80, 60), (120, 160), 2), ((2, 4, 56, 56), (160, 120), (120, 160), 1),
                                                     ((5, 3, 40, 52), (37, 29), (58, 74), 2), ((2, 11, 36, 36), (20, 15), (15, 20), 1)])
def test_random_score_maps_match_oracle(gpu, shape, win, frame, stride):
    """Smooth random score maps (blobs of several sizes, some classes empty, some saturated), resized up and down, windows laid out as
    detection_window_roi does: pmap and boxes bit for bit."""
    from scipy import ndimage as ndi
    rng = np.random.default_rng(sum(shape) + win[0])
    n, c, h, w = shape
    fm = np.stack([[ndi.gaussian_filter(rng.random((h, w)), rng.uniform(0.8, 3.0)) for _ in range(c)] for _ in range(n)]).astype(np.float32)
    fm = (fm - fm.min()) / (fm.max() - fm.min())
    fm[:, 1] = 0.2                       # a class entirely below the threshold
    if c > 3:
        fm[0, 3] = 1.0                   # a class that fires on every pixel
    rects = detection_window_roi((frame[0], frame[1], 3), stride)
    assert len(rects) == n and tuple(rects[0][2:]) == win
    for thr in (0.5, 0.62):
        got = score_masks_from_maps(fm, rects, frame, thr)
        want = M.run_detector2_post(fm, rects, frame, thr)
        found = same(got, want)
    assert found > 0 and got[0].max() > 0


def test_values_beyond_the_uint8_range_and_empty_maps(gpu):
    """Scores above 1 (x 255 > 255) wrap like ndarray.astype(np.uint8) does; all-zero maps give an empty pmap and no boxes."""
    rng = np.random.default_rng(3)
    fm = (rng.random((1, 3, 16, 16)) * 2.2).astype(np.float32)
    rects = [(0, 0, 24, 20)]
    assert same(score_masks_from_maps(fm, rects, (20, 24), 0.5), M.run_detector2_post(fm, rects, (20, 24), 0.5)) >= 0
    z = np.zeros((2, 3, 8, 8), np.float32)
    pmap, boxes = score_masks_from_maps(z, [(0, 0, 8, 8), (4, 4, 8, 8)], (12, 12), 0.5)
    assert pmap.max() == 0 and boxes == []


def test_run_detector2_masks_through_the_engine(gpu):
    """The node's path end to end: frame -> whole-frame normalisation -> windows (stride 1: the frame and the central crop) -> one
    batched forward of the VGG16-FCN deploy net -> the device's mask leg on its class-probability blob == the oracle on the blob the
    GPU produced."""
    msg = proto.parse_text(models.vgg16_fcn_bbox_deploy(2, 96, 128, 5))
    spec = NetSpec(msg, "TEST"); spec.infer()
    eng = Engine(NetSpec(msg, "TEST"), params=fill_params(spec, seed=5), device=0, autotune=False)
    det = FCNObjectDetector(eng, 0.2, 3, 0.2, HeadMapping.auto(list(eng.blobs)))      # 5 classes: softmax scores around 0.2
    rng = np.random.default_rng(11)
    frame = rng.integers(0, 256, (75, 101, 3), dtype=np.uint8)
    pmap, boxes = det.run_detector2_masks(frame, stride=1, score_blob="pool_score")
    score = eng.read_blob("pool_score")
    rects = detection_window_roi(frame.shape, 1)
    want = M.run_detector2_post(score, rects, frame.shape[:2], np.float32(0.2))
    assert same((pmap, boxes), want) > 0 and pmap.max() > 0
    with pytest.raises(ValueError):
        det.run_detector2_masks(frame, stride=2, score_blob="pool_score")
    eng.close()


def test_pycaffe_front_end_score_masks(gpu, tmp_path):
    """The node-side form: net.forward() then net.score_masks(rects, frame shape, threshold) == the oracle on net.blobs[...].data."""
    import sys
    from conftest import PYCAFFE
    if PYCAFFE not in sys.path:
        sys.path.insert(0, PYCAFFE)
    import caffe
    path = str(tmp_path / "deploy.prototxt")
    with open(path, "w") as f:
        f.write(models.vgg16_fcn_bbox_deploy(2, 64, 96, 4))
    caffe.set_device(0); caffe.set_mode_gpu()
    net = caffe.Net(path, caffe.TEST)
    net.blobs["data"].data[...] = np.random.default_rng(2).random((2, 3, 64, 96))
    net.forward()
    rects = detection_window_roi((50, 70, 3), 1)
    pmap, boxes = net.score_masks(rects, (50, 70), 0.25, score_blob="pool_score")
    want = M.run_detector2_post(net.blobs["pool_score"].data, rects, (50, 70), np.float32(0.25))
    assert same((pmap, boxes), want) >= 0 and pmap.shape == (50, 70)
