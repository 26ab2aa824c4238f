"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle):
CPU tests keep the oracle from drifting away from them; -m gpu tests hold the HIP path to the same vectors."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_err
from fcn_object_detector_amd import models, proto
from fcn_object_detector_amd.netspec import NetSpec, fill_params
from oracle import caffe_ref as R
from oracle import detect_ref as D
from oracle.net_ref import RefNet


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def test_oracle_reproduces_layer_fixtures():
    g = load("layers")
    assert np.allclose(R.relu(R.conv2d(g["conv_x"], g["conv_w"], g["conv_b"], 1, 1)), g["conv_y"], rtol=1e-5, atol=1e-6)
    assert np.allclose(R.conv2d(g["conv7_x"], g["conv7_w"], None, 3, 2), g["conv7_y"], rtol=1e-5, atol=1e-4)
    y, idx = R.max_pool(g["pool_x"], 3, 2, 0, return_index=True)
    assert np.array_equal(y, g["pool_y"]) and np.array_equal(idx, g["pool_idx"])
    assert np.array_equal(R.max_pool(g["pool_x"], 3, 1, 1), g["pool31_y"])
    assert np.allclose(R.lrn_across(g["lrn_x"], 5, 1e-4, 0.75, 1.0), g["lrn_y"], rtol=1e-6)
    assert np.allclose(R.deconv2d(g["deconv_x"], R.bilinear_filler((4, 1, 8, 8)), None, 2, 4, group=4), g["deconv_y"], rtol=1e-6, atol=1e-6)


def test_oracle_reproduces_net_detect_target_fixtures():
    g = load("net_64x96")
    assert np.array_equal(D.preprocess_frame(g["frame"], 96, 64)[None], g["data"])
    msg = proto.parse_text(models.googlenet_detectnet_deploy(1, 64, 96, 2))
    spec = NetSpec(msg, "TEST"); spec.infer()
    ref = RefNet(msg, "TEST", fill_params(spec, seed=1234))
    ref.blobs["data"] = g["data"]
    b = ref.forward()
    assert rel_err(b["coverage"], g["coverage"]) < 1e-5 and rel_err(b["bboxes"], g["bboxes"]) < 1e-5
    d = load("detect")
    det, lab = D.detect(d["cvg"][0], d["bbox"][0], 448, 448, 16, 0.5, 3, 0.2)
    assert np.array_equal(det, d["det"]) and np.array_equal(lab, d["lab"]) and len(det) >= 3
    det, lab = D.detect(d["cvg"][0], d["bbox"][0], 448, 448, 16, 0.5, 3, 0.2, round_mode="trunc")
    assert np.array_equal(det, d["det_trunc"]) and np.array_equal(lab, d["lab_trunc"])
    t = load("targets")
    offs = t["offsets"]
    for i in range(3):
        rects = [tuple(int(v) for v in r) for r in t["rects"][offs[i]:offs[i + 1]]]
        labels = [int(v) for v in t["labels"][offs[i]:offs[i + 1]]]
        out = D.bounding_box_parameterized_labels(448, 448, rects, labels, 16, 3)
        for name, o in zip(("fg", "bbox", "size", "obj", "cvg"), out):
            assert np.array_equal(o.astype(np.float32), t[name][i])


@pytest.mark.gpu
def test_hip_matches_layer_fixtures(gpu):
    from fcn_object_detector_amd import lib as L
    from gpu_util import conv_desc, dev_from, dev_to, nchw, nhwc, pack_ohwi
    g = load("layers")
    x, w, b = g["conv_x"], g["conv_w"], g["conv_b"]
    xd, wd, bd = dev_from(nhwc(x)), dev_from(pack_ohwi(w)), dev_from(b)
    yd = dev_from(np.zeros((1, 9, 11, 20), np.float32))
    d = conv_desc(xd, wd, bd, yd, 1, 9, 11, 12, 12, 20, 3, 1, 1, 9, 11, 20, 0, L.CONV_RELU)
    L.call("fcn_conv2d_fwd_f32", C.byref(d), None)
    assert rel_err(nchw(dev_to(yd, (1, 9, 11, 20)), 20), g["conv_y"]) < 1e-4
    x7, w7 = g["conv7_x"], g["conv7_w"]
    xd, wd = dev_from(nhwc(x7, 4)), dev_from(pack_ohwi(w7))
    yd = dev_from(np.zeros((1, 11, 9, 8), np.float32))
    d = conv_desc(xd, wd, None, yd, 1, 21, 17, 4, 4, 8, 7, 3, 2, 11, 9, 8)
    L.call("fcn_conv2d_fwd_f32", C.byref(d), None)
    assert rel_err(nchw(dev_to(yd, (1, 11, 9, 8)), 8), g["conv7_y"]) < 1e-4
    p = g["pool_x"]
    xd = dev_from(nhwc(p))
    yd, idd = dev_from(np.zeros((2, 7, 7, 8), np.float32)), dev_from(np.zeros((2, 7, 7, 8), np.int32))
    L.call("fcn_maxpool_fwd_f32", xd.ptr, yd.ptr, idd.ptr, 2, 15, 14, 8, 8, 3, 2, 0, 7, 7, 8, 0, None)
    assert np.array_equal(nchw(dev_to(yd, (2, 7, 7, 8)), 8), g["pool_y"])
    assert np.array_equal(dev_to(idd, (2, 7, 7, 8), np.int32).transpose(0, 3, 1, 2), g["pool_idx"])
    l = g["lrn_x"]
    xd, yd = dev_from(nhwc(l)), dev_from(np.zeros((1, 5, 6, 16), np.float32))
    L.call("fcn_lrn_fwd_f32", xd.ptr, yd.ptr, None, 30, 16, 16, 16, 5, 1e-4, 0.75, 1.0, None)
    assert rel_err(nchw(dev_to(yd, (1, 5, 6, 16)), 16), g["lrn_y"]) < 1e-5


@pytest.mark.gpu
def test_hip_matches_net_detect_target_fixtures(gpu):
    from fcn_object_detector_amd import lib as L
    from fcn_object_detector_amd.detector import FCNObjectDetector, HeadMapping, detect_from_maps, generate_targets
    from fcn_object_detector_amd.engine import Engine
    g = load("net_64x96")
    msg = proto.parse_text(models.googlenet_detectnet_deploy(1, 64, 96, 2))
    spec = NetSpec(msg, "TEST"); spec.infer()
    eng = Engine(NetSpec(msg, "TEST"), params=fill_params(spec, seed=1234), device=0)
    FCNObjectDetector(eng, mapping=HeadMapping.detectnet_deploy()).run_detector(g["frame"])      # device pre-processing
    assert np.abs(eng.read_blob("data") - g["data"]).max() <= 4e-6
    assert rel_err(eng.read_blob("coverage"), g["coverage"]) < 1e-3 and rel_err(eng.read_blob("bboxes"), g["bboxes"]) < 1e-3
    assert rel_err(eng.read_blob("pool3/3x3_s2")[:, :16], g["pool3"]) < 1e-3
    assert abs(float(eng.read_blob("inception_4c/output").astype(np.float64).sum()) - g["inc4c_sum"][0]) < 1e-3 * abs(g["inc4c_sum"][0])
    eng.close()
    d = load("detect")
    det, lab = detect_from_maps(d["cvg"], d["bbox"], 448, 448)[0]
    assert np.array_equal(det, d["det"]) and np.array_equal(lab, d["lab"])
    det, lab = detect_from_maps(d["cvg"], d["bbox"], 448, 448, round_mode=L.RECT_ROUND_TRUNCATE)[0]
    assert np.array_equal(det, d["det_trunc"]) and np.array_equal(lab, d["lab_trunc"])
    t = load("targets")
    offs = t["offsets"]
    rects = [[tuple(int(v) for v in r) for r in t["rects"][offs[i]:offs[i + 1]]] for i in range(3)]
    labels = [[int(v) for v in t["labels"][offs[i]:offs[i + 1]]] for i in range(3)]
    out = generate_targets(rects, labels, 448, 448, 16, 3)
    for name, o in zip(("fg", "bbox", "size", "obj", "cvg"), out):
        assert np.array_equal(o, t[name])


def _train_setup():
    import importlib.util
    spec_ = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mg)
    msg = proto.parse_text(models.googlenet_detectnet_train("m", "L", "unused", num_classes=1))
    data = mg.train_batch()
    spec = NetSpec(msg, "TRAIN")
    spec.infer({k: v.shape for k, v in data.items()})
    return msg, spec, data, fill_params(spec, seed=4321)


TRAIN_LAYERS = ("conv1/7x7_s2", "inception_4a/1x1", "bbox/regressor")


def test_oracle_reproduces_training_fixture():
    """One solver iteration of the DetectNet training net: losses, gradients, updated weights (tests/golden/train_64x96.npz)."""
    from oracle.net_ref import RefSolver
    g = load("train_64x96")
    msg, spec, data, params = _train_setup()
    ref = RefNet(msg, "TRAIN", {k: [a.copy() for a in v] for k, v in params.items()})
    ref.blobs.update(data)
    ref.dropout_seed = 11
    ref.forward()
    assert abs(ref.losses["loss_bbox"] - g["loss_bbox"][0]) < 1e-6 * abs(g["loss_bbox"][0])
    assert abs(ref.total_loss() - g["total"][0]) < 1e-6 * abs(g["total"][0])
    grads = ref.backward()
    assert rel_err(ref.diffs["bboxes"], g["d_bboxes"]) < 1e-6 and rel_err(ref.diffs["pool5/drop_s1"], g["d_pool5"]) < 1e-5
    smsg = proto.parse_text('base_lr: 0.001 momentum: 0.9 weight_decay: 1e-6 lr_policy: "fixed"')
    RefSolver(ref, smsg, {l.name: l.lr_mult for l in spec.param_layers()}, {l.name: l.decay_mult for l in spec.param_layers()}).apply(grads)
    for name in TRAIN_LAYERS:
        key = name.replace("/", "__")
        assert rel_err(grads[name][0], g["dw_" + key]) < 1e-5 and rel_err(grads[name][1], g["db_" + key]) < 1e-5
        assert rel_err(ref.params[name][0], g["w_after_" + key]) < 1e-6


@pytest.mark.gpu
def test_hip_matches_training_fixture(gpu):
    from fcn_object_detector_amd.train import SolverParams, TrainEngine
    g = load("train_64x96")
    msg, spec, data, params = _train_setup()
    eng = TrainEngine(NetSpec(msg, "TRAIN"), {k: v.shape for k, v in data.items()}, params={k: [a.copy() for a in v] for k, v in params.items()},
                      device=0, solver=SolverParams(base_lr=1e-3, momentum=0.9, weight_decay=1e-6, lr_policy="fixed"), autotune=False)
    for k, v in data.items():
        eng.host_array(k)[...] = v
    out = eng.step(seed=11)
    assert abs(out["loss_bbox"] - g["loss_bbox"][0]) < 1e-3 * abs(g["loss_bbox"][0])
    assert abs(out["loss_coverage"] - g["loss_coverage"][0]) < 1e-3 * abs(g["loss_coverage"][0])
    assert abs(out["total_loss"] - g["total"][0]) < 1e-3 * abs(g["total"][0])
    assert rel_err(eng.read_blob("coverage"), g["coverage"]) < 1e-3 and rel_err(eng.read_blob("bboxes"), g["bboxes"]) < 1e-3
    assert rel_err(eng.read_grad("bboxes"), g["d_bboxes"]) < 1e-3
    got, after = eng.download_grads(), eng.download_params()
    for name in TRAIN_LAYERS:
        key = name.replace("/", "__")
        assert rel_err(after[name][0], g["w_after_" + key]) < 1e-3, name
    # weight gradients: against the oracle's backward evaluated on the device's own forward pass (identical ReLU masks and
    # pooling argmaxes - gpu_util.adopt_device_activations), at the north-star tolerance
    from gpu_util import adopt_device_activations
    ref = RefNet(msg, "TRAIN", {k: [a.copy() for a in v] for k, v in params.items()})
    ref.blobs.update(data)
    ref.dropout_seed = 11
    ref.forward()
    adopt_device_activations(ref, eng, spec, keep=data)
    grads = ref.backward()
    for name in TRAIN_LAYERS:
        for gg, rr in zip(got[name], grads[name]):
            assert rel_err(gg, rr) < 1e-3, name
    eng.close()


# ---- vectors computed by the REFERENCE's own functions (tests/golden/make_reference_golden.py; CPU twins in
# tests/test_reference_golden.py): SURVEY.md rows A4 and A7 on the device ------------------------------------------------

def _ref_rows():
    return np.load(os.path.join(GOLDEN, "reference_numpy_rows.npz"))


@pytest.mark.gpu
def test_hip_target_generation_matches_reference_vectors(gpu):
    """fcn_gen_targets against what argumentation_engine.py:69-109 itself produced (float64 there, float32 tops here)."""
    from fcn_object_detector_amd.detector import generate_targets
    G = _ref_rows()
    n = 0
    for name in (str(v) for v in G["a4_names"]):
        h, w, s, c = (int(v) for v in G["a4_%s_meta" % name])
        rects = [tuple(int(v) for v in r) for r in G["a4_%s_rects" % name]]
        labels = [int(v) for v in G["a4_%s_labels" % name]]
        got = generate_targets([rects], [labels], w, h, s, c)
        for key, arr in zip(("fg", "bbox", "size", "obj", "cvg"), got):
            with np.errstate(all="ignore"):
                ref = G["a4_%s_%s" % (name, key)].astype(np.float32)      # the layer stores into float32 tops (:113-121)
            assert arr[0].shape == ref.shape and np.array_equal(arr[0], ref, equal_nan=True), (name, key)
        n += 1
    assert n >= 10


@pytest.mark.gpu
def test_hip_decode_matches_reference_gridbox_vectors(gpu):
    """The decode half of fcn_detect_decode_group against the reference's gridbox_to_boxes output (fcn_object_detector.py:357-394
    at stride 8, boundary_refinement.py:265-302 at stride 16): with groupThreshold 0 cv::groupRectangles returns its input, so the
    kernel's output is the candidate list itself - the reference's float64 boxes through the cv2 Rect converter, in np.where order."""
    from fcn_object_detector_amd.detector import detect_from_maps
    G = _ref_rows()
    i = checked = 0
    while "a7_%d_s8_meta" % i in G:
        for tag in ("s8", "s16"):
            key = "a7_%d_%s" % (i, tag)
            net_w, net_h, stride = (int(v) for v in G[key + "_meta"])
            cvg, bb, boxes = G[key + "_cvg"], G[key + "_bbox"], G[key + "_boxes"]
            for mode, lmode in (("nearest_even", 0), ("trunc", 1)):
                dets, labels = detect_from_maps(cvg[None, None], bb[None], net_w, net_h, float(G[key + "_thresh"]), 0, 0.2,
                                                min_height=-(1 << 30), round_mode=lmode)[0]
                want = np.array([D.to_rect(b, mode) for b in boxes.tolist()], np.float64).reshape(-1, 4) if boxes.any() else np.zeros((0, 4))
                assert dets.shape == (len(want), 5) and np.array_equal(dets[:, :4], want), (key, mode)
                assert not dets[:, 4].any() and not labels.any()      # log(weight 1) = 0, class 0
                checked += len(want)
        i += 1
    assert i == 5 and checked > 6000      # (640 x 480 at stride 8 - 4800 cells - included since round 4)
