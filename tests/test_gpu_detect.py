"""Bit-exact GPU parity of decode + groupRectangles, target generation and pre-processing vs the oracle (-m gpu)."""
import numpy as np
import pytest

from fcn_object_detector_amd import lib as L
from fcn_object_detector_amd import models, proto
from fcn_object_detector_amd.detector import FCNObjectDetector, HeadMapping, detect_from_maps, generate_targets
from fcn_object_detector_amd.engine import DeviceBuffer, Engine
from fcn_object_detector_amd.netspec import NetSpec, fill_params
from oracle import detect_ref as D

pytestmark = pytest.mark.gpu


def synth_maps(rng, n, c, gy, gx, stride, n_obj=3, noise=0.08):
    """Coverage blobs around a few objects + box regressions pointing at them with jitter (what a trained head emits)."""
    cvg = (rng.random((n, c, gy, gx)) * noise).astype(np.float32)
    bb = (rng.standard_normal((n, 4 * c, gy, gx)) * 2).astype(np.float32)
    for i in range(n):
        for _ in range(n_obj):
            k = int(rng.integers(0, c))
            w, h = rng.integers(3 * stride, 9 * stride, 2)
            x0 = int(rng.integers(0, gx * stride - w)); y0 = int(rng.integers(0, gy * stride - h))
            for cy in range(y0 // stride, min((y0 + h) // stride + 1, gy)):
                for cx in range(x0 // stride, min((x0 + w) // stride + 1, gx)):
                    cvg[i, k, cy, cx] = 0.5 + 0.5 * rng.random()
                    j = rng.standard_normal(4) * 2.5
                    bb[i, 4 * k:4 * k + 4, cy, cx] = [x0 - cx * stride + j[0], y0 - cy * stride + j[1],
                                                     x0 + w - cx * stride + j[2], y0 + h - cy * stride + j[3]]
    return cvg, bb


def check(cvg, bb, im, stride, **kw):
    got = detect_from_maps(cvg, bb, im, im, **{k: v for k, v in kw.items() if k != "mode"},
                           round_mode=L.RECT_ROUND_TRUNCATE if kw.get("mode") == "trunc" else L.RECT_ROUND_NEAREST_EVEN)
    total = 0
    for i in range(len(cvg)):
        rdet, rlab = D.detect(cvg[i], bb[i], im, im, stride, kw.get("prob_thresh", 0.5), kw.get("min_boxes", 3), kw.get("eps", 0.2),
                              "trunc" if kw.get("mode") == "trunc" else "nearest_even", fast=True)
        assert np.array_equal(got[i][1], rlab)
        assert np.array_equal(got[i][0], rdet)          # ints and log(n) bit-for-bit
        total += len(rdet)
    return total


@pytest.mark.parametrize("c,grid,stride", [(4, 28, 16), (10, 56, 8), (1, 28, 16)])
def test_detect_random_scenes(gpu, c, grid, stride):
    rng = np.random.default_rng(100 + c)
    n_det = 0
    for trial in range(6):
        cvg, bb = synth_maps(rng, 2, c, grid, grid, stride, n_obj=4)
        n_det += check(cvg, bb, grid * stride, stride)
        n_det += check(cvg, bb, grid * stride, stride, mode="trunc")
    assert n_det > 10                      # the scenes really produce clusters


def test_detect_thresholds_and_eps(gpu):
    rng = np.random.default_rng(7)
    cvg, bb = synth_maps(rng, 1, 3, 28, 28, 16, n_obj=5)
    for kw in (dict(prob_thresh=0.3), dict(prob_thresh=0.75, min_boxes=1), dict(eps=0.05), dict(eps=0.6, min_boxes=2),
               dict(min_boxes=0), dict(min_boxes=10)):
        check(cvg, bb, 448, 16, **kw)


def test_detect_edge_cases(gpu):
    z = np.zeros((1, 2, 28, 28), np.float32)
    zb = np.zeros((1, 8, 28, 28), np.float32)
    assert check(z, zb, 448, 16) == 0                                   # nothing above threshold
    one = z.copy(); one[0, 0, 0, 0] = 1.0
    assert check(one, zb, 448, 16) == 0                                 # one candidate, all-zero box: `.any()` early-out
    full = np.ones((1, 1, 56, 56), np.float32)                          # all 3136 cells positive: worst case O(M^2)
    rng = np.random.default_rng(9)
    fb = (rng.standard_normal((1, 4, 56, 56)) * 6).astype(np.float32)
    fb[0, 2:] += 40
    check(full, fb, 448, 8)
    # half-way coordinates exercise round-half-even vs truncation
    half = np.zeros((1, 1, 28, 28), np.float32); hb = np.zeros((1, 4, 28, 28), np.float32)
    half[0, 0, 4, 4:9] = 0.9
    for i, cx in enumerate(range(4, 9)):
        hb[0, :, 4, cx] = [10.5 - 16 * i, 3.5, 90.5 - 16 * i, 61.5]
    assert check(half, hb, 448, 16) == 1 and check(half, hb, 448, 16, mode="trunc") == 1


def test_detect_grids_beyond_4096_cells(gpu):
    """Round 4: the kernel's bound is on a class's CANDIDATES (5120, the LDS), not on the grid.  640 x 480 at stride 8 - the node's
    camera frame on the stride-8 heads (fcn_object_detector.py:357-394), 4800 cells - with EVERY cell firing, in both rounding
    modes; a 6400-cell grid with ordinary scenes; and the same grid with every cell firing, which must be refused loudly."""
    rng = np.random.default_rng(11)
    gy, gx, stride = 60, 80, 8
    full = np.ones((1, 1, gy, gx), np.float32)
    fb = (rng.standard_normal((1, 4, gy, gx)) * 6).astype(np.float32)
    fb[0, 2:] += 40
    for mode, lmode in (("nearest_even", L.RECT_ROUND_NEAREST_EVEN), ("trunc", L.RECT_ROUND_TRUNCATE)):
        got = detect_from_maps(full, fb, gx * stride, gy * stride, round_mode=lmode)
        rdet, rlab = D.detect(full[0], fb[0], gx * stride, gy * stride, stride, 0.5, 3, 0.2, mode, fast=True)
        assert len(rdet) > 0 and np.array_equal(got[0][0], rdet) and np.array_equal(got[0][1], rlab)
    cvg, bb = synth_maps(rng, 2, 3, 80, 80, 8, n_obj=6)
    assert check(cvg, bb, 640, 8) > 0
    with pytest.raises(RuntimeError, match="5120"):
        detect_from_maps(np.ones((1, 1, 80, 80), np.float32), np.ones((1, 4, 80, 80), np.float32), 640, 640)


def test_targets_kats_and_random(gpu):
    fg, bl, sl, ol, cl = generate_targets([[(100, 120, 80, 60)]], [[0]], 448, 448, 16, 1)
    assert int(fg.sum()) == 28 and bl[0, :, 8, 7].tolist() == [-12.0, -8.0, 68.0, 52.0]
    out = generate_targets([[(40, 64, 120, 200), (300, 310, 64, 48)]], [[3, 10]], 448, 448, 8, 11)
    assert [int(out[0][0, c].sum()) for c in range(11)] == [0, 0, 0, 375, 0, 0, 0, 0, 0, 0, 63]
    assert out[1].sum() == 7260.0 and out[4].sum() == 1752
    rng = np.random.default_rng(42)
    for (im, stride, C, batch) in [(448, 16, 1, 8), (288, 8, 11, 4), (224, 16, 3, 5)]:
        rects, labels = [], []
        for b in range(batch):
            n = int(rng.integers(0, 4))                                    # includes images without boxes
            rs = []
            for _ in range(n):
                w, h = rng.integers(8, im // 2, 2)
                rs.append((int(rng.integers(-10, im - w + 10)), int(rng.integers(-10, im - h + 10)), int(w), int(h)))
            rects.append(rs); labels.append([int(v) for v in rng.integers(0, C, n)])
        got = generate_targets(rects, labels, im, im, stride, C)
        for b in range(batch):
            ref = D.bounding_box_parameterized_labels(im, im, rects[b], labels[b], stride, C)
            for g, r in zip(got, ref):
                assert np.array_equal(g[b], r.astype(np.float32))          # bit-exact incl. later-rect-overwrites


def test_targets_overlapping_rects_last_wins(gpu):
    rects = [[(100, 100, 120, 120), (130, 110, 100, 90)]]
    got = generate_targets(rects, [[0, 0]], 448, 448, 16, 1)
    ref = D.bounding_box_parameterized_labels(448, 448, rects[0], [0, 0], 16, 1)
    for g, r in zip(got, ref):
        assert np.array_equal(g[0], r.astype(np.float32))
    with pytest.raises(IndexError):
        generate_targets(rects, [[0, 5]], 448, 448, 16, 1)


@pytest.mark.parametrize("h,w", [(448, 448), (480, 640), (300, 517)])
def test_node_pipeline_matches_oracle(gpu, h, w):
    """uint8 frame -> device pre-processing -> forward -> device decode/group == oracle on the same frame."""
    msg = proto.parse_text(models.googlenet_detectnet_deploy(1, 448, 448, 4))
    spec = NetSpec(msg, "TEST"); spec.infer()
    params = fill_params(spec, seed=1234)
    eng = Engine(NetSpec(msg, "TEST"), params=params, device=0)
    det = FCNObjectDetector(eng, 0.5, 3, 0.2, HeadMapping.detectnet_deploy())
    frame = np.random.default_rng(h).integers(0, 256, (h, w, 3), dtype=np.uint8)
    boxes, labels = det.run_detector(frame)
    blob = D.preprocess_frame(frame, 448, 448)
    got = eng.read_blob("data")[0]
    # pre-processing parity (f64 math, f32 store).  The device keeps the Power(shift=-127)'d blob, like Caffe's
    # transformed_data; reading `data` back adds 127 again, so it is exact to half an f32 ulp at 127 (3.8e-6)
    assert np.abs(got - blob).max() <= 4e-6
    assert np.abs(eng.read_blob("transformed_data")[0] - (blob + np.float32(-127.0))).max() == 0.0
    # post-processing is checked bit-exactly on the maps the GPU produced
    cvg, bb = eng.read_blob("coverage")[0], eng.read_blob("bboxes")[0]
    rdet, rlab = D.detect(cvg, bb, 448, 448, 16, 0.5, 3, 0.2, fast=True)
    rbox = np.asarray(rdet, dtype=np.int64).reshape(-1, 5)
    if len(rbox):
        rbox = D.resize_detection(frame.shape, rbox, 448, 448)
    assert np.array_equal(boxes, rbox) and np.array_equal(labels, rlab)
    eng.close()


def test_batched_node_pipeline_matches_oracle(gpu):
    """BASELINE configs[4] without the fp16 arithmetic: a batch of frames of different sizes -> pre-processing -> one forward
    -> ONE fused decode + groupRectangles launch; per frame bit-equal to the oracle on the maps the GPU produced.  The head
    biases are raised so that the random-weight net really emits detections."""
    batch = 4
    msg = proto.parse_text(models.googlenet_detectnet_deploy(batch, 224, 320, 3))
    spec = NetSpec(msg, "TEST"); spec.infer()
    params = fill_params(spec, seed=77)
    rng = np.random.default_rng(5)
    params["cvg/classifier"][1][...] = 1.5                              # sigmoid > 0.5 on most cells
    params["bbox/regressor"][0][...] = 0                                # boxes = bias pattern: all cells vote for similar rects
    params["bbox/regressor"][1][...] = np.tile(np.array([-30, -25, 35, 40], np.float32), 3) + rng.normal(0, 0.5, 12).astype(np.float32)
    eng = Engine(NetSpec(msg, "TEST"), params=params, device=0, autotune=False)
    det = FCNObjectDetector(eng, 0.5, 3, 0.2, HeadMapping.detectnet_deploy())
    frames = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in ((224, 320), (480, 640), (100, 517), (300, 200))]
    res = det.run_detector_batch(frames)
    cvg, bb = eng.read_blob("coverage"), eng.read_blob("bboxes")
    total = 0
    for i, (frame, (boxes, labels)) in enumerate(zip(frames, res)):
        rdet, rlab = D.detect(cvg[i], bb[i], 320, 224, 16, 0.5, 3, 0.2, fast=True)
        rbox = np.asarray(rdet, dtype=np.int64).reshape(-1, 5)
        if len(rbox):
            rbox = D.resize_detection(frame.shape, rbox, 320, 224)
        assert np.array_equal(boxes, rbox) and np.array_equal(labels, rlab), i
        assert np.abs(eng.read_blob("data")[i] - D.preprocess_frame(frame, 320, 224)).max() <= 4e-6
        total += len(boxes)
    assert total > 0
    with pytest.raises(ValueError):
        det.run_detector_batch(frames[:2])
    # equally sized frames take the three-launch batched pre-processing: same blob as frame by frame
    same = [rng.integers(0, 256, (240, 352, 3), dtype=np.uint8) for _ in range(batch)]
    det.run_detector_batch(same)
    for i, f in enumerate(same):
        assert np.abs(eng.read_blob("data")[i] - D.preprocess_frame(f, 320, 224)).max() <= 4e-6
    eng.close()


def test_detector_pipeline_equals_frame_by_frame(gpu):
    """DetectorPipeline (three replica engines, frames of different sizes in flight together): per frame exactly what
    run_detector returns, in frame order; the head biases are raised so that there are detections to compare."""
    from fcn_object_detector_amd.detector import DetectorPipeline
    msg = proto.parse_text(models.googlenet_detectnet_deploy(1, 224, 320, 3))
    spec = NetSpec(msg, "TEST"); spec.infer()
    params = fill_params(spec, seed=77)
    rng = np.random.default_rng(6)
    params["cvg/classifier"][1][...] = 1.5
    params["bbox/regressor"][0][...] = 0
    params["bbox/regressor"][1][...] = np.tile(np.array([-30, -25, 35, 40], np.float32), 3) + rng.normal(0, 0.5, 12).astype(np.float32)
    mapping = HeadMapping.detectnet_deploy()
    lone = FCNObjectDetector(Engine(NetSpec(msg, "TEST"), params=params, device=0, autotune=False), 0.5, 3, 0.2, mapping)
    pipe = DetectorPipeline(lambda first: Engine(NetSpec(msg, "TEST"), params=params, device=0, autotune=False, tune_from=first), depth=3,
                            detection_threshold=0.5, min_boxes=3, nms_eps=0.2, mapping=mapping)
    sizes = [(224, 320), (480, 640), (100, 517), (300, 200), (480, 640), (224, 320), (97, 131)]
    frames = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    want = [lone.run_detector(f) for f in frames]
    got = pipe.run_detector_stream(frames)
    assert len(got) == len(frames) and sum(len(b) for b, _ in want) > 0
    for (gb, gl), (wb, wl) in zip(got, want):
        assert np.array_equal(gb, wb) and np.array_equal(gl, wl)
    pipe.submit(frames[0])
    with pytest.raises(RuntimeError):
        pipe._queue[0].submit(frames[1])                # one frame per replica
    assert np.array_equal(pipe.collect()[0], want[0][0])
    with pytest.raises(RuntimeError):
        pipe.collect()
    pipe.close()
    lone.engine.close()


def test_batched_detector_pipeline(gpu):
    """Two batch-4 replicas, three batches in a row: each batch's result equals run_detector_batch on a lone detector."""
    from fcn_object_detector_amd.detector import DetectorPipeline
    batch = 4
    msg = proto.parse_text(models.googlenet_detectnet_deploy(batch, 224, 320, 3))
    spec = NetSpec(msg, "TEST"); spec.infer()
    params = fill_params(spec, seed=77)
    rng = np.random.default_rng(8)
    params["cvg/classifier"][1][...] = 1.5
    params["bbox/regressor"][0][...] = 0
    params["bbox/regressor"][1][...] = np.tile(np.array([-30, -25, 35, 40], np.float32), 3) + rng.normal(0, 0.5, 12).astype(np.float32)
    mapping = HeadMapping.detectnet_deploy()
    lone = FCNObjectDetector(Engine(NetSpec(msg, "TEST"), params=params, device=0, autotune=False), 0.5, 3, 0.2, mapping)
    pipe = DetectorPipeline(lambda first: Engine(NetSpec(msg, "TEST"), params=params, device=0, autotune=False, tune_from=first), depth=2,
                            mapping=mapping)
    batches = [[rng.integers(0, 256, (240, 352, 3), dtype=np.uint8) for _ in range(batch)],
               [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in ((224, 320), (480, 640), (100, 517), (300, 200))],
               [rng.integers(0, 256, (224, 320, 3), dtype=np.uint8) for _ in range(batch)]]
    want = [lone.run_detector_batch(b) for b in batches]
    got = pipe.run_detector_batches(batches)
    assert sum(len(bx) for res in want for bx, _ in res) > 0
    for g, w in zip(got, want):
        for (gb, gl), (wb, wl) in zip(g, w):
            assert np.array_equal(gb, wb) and np.array_equal(gl, wl)
    pipe.close()
    lone.engine.close()


@pytest.mark.parametrize("grid,stride,c", [(28, 16, 2), (56, 8, 1)])
def test_detect_every_cell_fires(gpu, grid, stride, c):
    """Worst case of the clustering kernel: every cell is a candidate (M = 784 / 3136, hundreds of thousands of
    SimilarRects tests, dense and sparse neighbourhoods), as an untrained net produces."""
    rng = np.random.default_rng(grid)
    cvg = (0.5 + 0.5 * rng.random((1, c, grid, grid))).astype(np.float32)
    bb = (rng.standard_normal((1, 4 * c, grid, grid)) * 6).astype(np.float32)            # boxes ~ cell origin: many chains of neighbours
    assert check(cvg, bb, grid * stride, stride) >= 0
    bb2 = bb.copy()
    bb2[0, 2::4] += 60
    bb2[0, 3::4] += 45                                                                     # a real extent: clusters of overlapping votes
    assert check(cvg, bb2, grid * stride, stride, min_boxes=2) > 0
    assert check(cvg, bb2, grid * stride, stride, eps=1.5) >= 0                            # everything similar to everything nearby


def test_detect_sizes_outside_the_margin_table(gpu):
    """min(w) + min(h) beyond the kernel's 4096-entry floor(delta) table and negative sizes take the arithmetic path."""
    rng = np.random.default_rng(3)
    cvg = np.zeros((1, 1, 28, 28), np.float32)
    bb = np.zeros((1, 4, 28, 28), np.float32)
    for k, (cy, cx) in enumerate([(2, 3), (2, 4), (3, 3), (3, 4), (2, 5), (10, 10), (10, 11), (11, 10), (11, 11), (12, 12)]):
        cvg[0, 0, cy, cx] = 0.9
        big = k < 5
        base = np.array([40, 30, 5200, 4700], np.float32) if big else np.array([-300, -200, -90, -70], np.float32)
        bb[0, :, cy, cx] = base - np.array([cx * 16, cy * 16, cx * 16, cy * 16], np.float32) + rng.integers(-3, 4, 4)
    assert check(cvg, bb, 448, 16, min_boxes=2) >= 1          # the large boxes group; negative sizes never do (delta < 0)


@pytest.mark.parametrize("h,w,stride", [(480, 640, 2), (301, 517, 2), (240, 330, 1), (448, 448, 3)])
def test_run_detector2_windows_match_oracle(gpu, h, w, stride):
    """run_detector2's batch (fcn_object_detector.py:178-211, :257-277): stride^2 windows + the central one, cut from the frame
    AFTER whole-frame normalisation, resized to the net input, one batched forward, one decode + groupRectangles launch.  Window
    geometry bit-exact; blob contents as the single-frame path (f64 arithmetic, one f32 rounding); boxes bit-exact on the
    maps the GPU produced, in frame coordinates."""
    batch = stride * stride + 1
    msg = proto.parse_text(models.googlenet_detectnet_deploy(batch, 160, 192, 2))
    spec = NetSpec(msg, "TEST"); spec.infer()
    params = fill_params(spec, seed=3)
    rng = np.random.default_rng(h + stride)
    params["cvg/classifier"][1][...] = 1.5
    params["bbox/regressor"][0][...] = 0
    params["bbox/regressor"][1][...] = np.tile(np.array([-30, -25, 35, 40], np.float32), 2) + rng.normal(0, 0.5, 8).astype(np.float32)
    eng = Engine(NetSpec(msg, "TEST"), params=params, device=0, autotune=False)
    det = FCNObjectDetector(eng, 0.5, 3, 0.2, HeadMapping.detectnet_deploy())
    frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    rects, res = det.run_detector2(frame, stride)
    want, rrects = D.run_detector2_inputs(frame, 192, 160, stride)
    assert np.array_equal(rects, np.asarray(rrects, dtype=np.int32))
    assert np.abs(eng.read_blob("data") - want).max() <= 4e-6
    cvg, bb = eng.read_blob("coverage"), eng.read_blob("bboxes")
    total = 0
    for i, (boxes, labels) in enumerate(res):
        rdet, rlab = D.detect(cvg[i], bb[i], 192, 160, 16, 0.5, 3, 0.2, fast=True)
        assert np.array_equal(boxes, D.window_boxes_to_frame(rrects[i], rdet, 192, 160)) and np.array_equal(labels, rlab), i
        total += len(boxes)
    assert total > 0
    with pytest.raises(ValueError):
        det.run_detector2(frame, stride + 1)                  # window count != the engine's batch
    eng.close()


def test_run_detector2_on_the_half_float_engine(gpu):
    """The window batch through the f16 engine: the windows are written as whole 8-half pixels (dst_f16 = 3: b, g, r and the two constant
    channels of the folded Power shift), the first layer takes the constant-channel kernel; blob within half an f16 ulp of the oracle,
    boxes bit-exact on the maps the GPU produced."""
    msg = proto.parse_text(models.googlenet_detectnet_deploy(5, 160, 192, 2))
    spec = NetSpec(msg, "TEST"); spec.infer()
    params = fill_params(spec, seed=3)
    rng = np.random.default_rng(21)
    params["cvg/classifier"][1][...] = 1.5
    params["bbox/regressor"][0][...] = 0
    params["bbox/regressor"][1][...] = np.tile(np.array([-30, -25, 35, 40], np.float32), 2) + rng.normal(0, 0.5, 8).astype(np.float32)
    eng = Engine(NetSpec(msg, "TEST"), params=params, device=0, autotune=False, dtype="f16")
    det = FCNObjectDetector(eng, 0.5, 3, 0.2, HeadMapping.detectnet_deploy())
    frame = rng.integers(0, 256, (301, 517, 3), dtype=np.uint8)
    rects, res = det.run_detector2(frame, 2)
    want, rrects = D.run_detector2_inputs(frame, 192, 160, 2)
    assert np.array_equal(rects, np.asarray(rrects, dtype=np.int32))
    assert np.abs(eng.read_blob("data") - want).max() <= 2.0 ** -11      # [0, 1] values rounded once to halves
    cvg, bb = eng.read_blob("coverage"), eng.read_blob("bboxes")
    total = 0
    for i, (boxes, labels) in enumerate(res):
        rdet, rlab = D.detect(cvg[i], bb[i], 192, 160, 16, 0.5, 3, 0.2, fast=True)
        assert np.array_equal(boxes, D.window_boxes_to_frame(rrects[i], rdet, 192, 160)) and np.array_equal(labels, rlab), i
        total += len(boxes)
    assert total > 0
    eng.close()


def test_roi_preprocessing_refuses_windows_outside_the_frame(gpu):
    lib = L.load()
    L.call("fcn_init", 0)
    frame, dst, mm = DeviceBuffer(64 * 48 * 3), DeviceBuffer(2 * 32 * 32 * 4 * 4), DeviceBuffer(32)
    for bad in ([0, 0, 65, 10], [-1, 0, 10, 10], [10, 40, 10, 9], [0, 0, 0, 5]):
        rois = np.asarray([[0, 0, 64, 48], bad], np.int32)
        assert lib.fcn_preprocess_bgr8_rois(frame.ptr, 48, 64, rois.ctypes.data, 2, dst.ptr, 0, 32, 32, 4, 0.0, mm.ptr, None) == 1      # FCN_E_ARG
    too_many = np.tile(np.asarray([[0, 0, 8, 8]], np.int32), (33, 1))
    assert lib.fcn_preprocess_bgr8_rois(frame.ptr, 48, 64, too_many.ctypes.data, 33, dst.ptr, 0, 32, 32, 4, 0.0, mm.ptr, None) == 1      # FCN_E_ARG


def test_preprocessing_writes_whole_half_pixels_when_asked(gpu):
    """dst_f16 = 3 (the f16 engine's half image): every pixel leaves as (b, g, r, 1, 1, 0, 0, 0) in one store - channels 0..2 bit-equal to
    the three-store form (dst_f16 = 1), which leaves the other channels alone; refused for other pixel widths."""
    lib = L.load()
    L.call("fcn_init", 0)
    rng = np.random.default_rng(11)
    n, h, w, H, W = 3, 37, 53, 24, 40
    frames = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    src = DeviceBuffer(frames.nbytes, zero=False)
    L.call("fcn_memcpy_h2d_async", src.ptr, frames.ctypes.data, frames.nbytes, None)
    mm = DeviceBuffer(32 * n)
    outs = []
    for flag in (1, 3):
        fill = np.full((n, H, W, 8), -7.0, np.float16)
        dst = DeviceBuffer(fill.nbytes, zero=False)
        L.call("fcn_memcpy_h2d_async", dst.ptr, fill.ctypes.data, fill.nbytes, None)
        L.call("fcn_preprocess_bgr8_batch", src.ptr, n, h, w, dst.ptr, flag, H, W, 8, 0.0, mm.ptr, None)
        got = np.empty_like(fill)
        L.call("fcn_memcpy_d2h_async", got.ctypes.data, dst.ptr, got.nbytes, None)
        L.call("fcn_device_sync")
        outs.append(got)
    assert np.array_equal(outs[0][..., :3], outs[1][..., :3])
    assert np.all(outs[0][..., 3:] == np.float16(-7.0))
    assert np.all(outs[1][..., 3:5] == np.float16(1.0)) and np.all(outs[1][..., 5:] == 0)
    for i in range(n):      # and both equal the oracle's blob, rounded once to half
        want = D.preprocess_frame(frames[i], W, H).transpose(1, 2, 0).astype(np.float16)
        assert np.array_equal(outs[1][i, ..., :3], want)
    dst4 = DeviceBuffer(n * H * W * 4 * 2)
    assert lib.fcn_preprocess_bgr8_batch(src.ptr, n, h, w, dst4.ptr, 3, H, W, 4, 0.0, mm.ptr, None) != 0
