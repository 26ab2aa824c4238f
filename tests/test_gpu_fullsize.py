"""BASELINE.json configs[2] and configs[4] at their FULL sizes against the oracle (-m gpu).

configs[2]: one training step of the DetectNet GoogLeNet net at batch 8, 448x448 - label tensors generated on the device
from synthetic boxes (bit-exact), forward losses against the oracle's own forward (1e-3), gradients of the heads and of the
top of the net against the oracle's backward on the device's activations (1e-3; gpu_util.adopt_device_activations).
configs[4]: batch 32, 448x448, half-float engine + ONE fused decode / groupRectangles launch - the f16 maps against the f32
engine, and the detections bit-exact against the integer oracle on the very maps the GPU produced, with head biases raised
so that the random-weight net emits detections (every class has candidates; the counts are asserted)."""
import numpy as np
import pytest

from conftest import elem_err, rel_err
from fcn_object_detector_amd import models, proto
from fcn_object_detector_amd.detector import FCNObjectDetector, HeadMapping
from fcn_object_detector_amd.engine import Engine
from fcn_object_detector_amd.netspec import NetSpec, fill_params
from fcn_object_detector_amd.train import SolverParams, TrainEngine
from gpu_util import adopt_device_activations
from oracle import detect_ref as D
from oracle.net_ref import RefNet

pytestmark = pytest.mark.gpu

LABELS = ("coverage-label", "bbox-label", "size-block", "obj-block", "coverage-block")


def synth_boxes(rng, n, size=448):
    """per image 1-3 rects, w, h ~ U{32..224}, x, y ~ U{0..size-1-w}, label 0."""
    rects = []
    for _ in range(n):
        rs = []
        for _ in range(int(rng.integers(1, 4))):
            w, h = int(rng.integers(32, 225)), int(rng.integers(32, 225))
            rs.append((int(rng.integers(0, size - w)), int(rng.integers(0, size - h)), w, h))
        rects.append(rs)
    return rects


def test_config2_train_step_batch8_448(gpu):
    n, size = 8, 448
    msg = proto.parse_text(models.googlenet_detectnet_train("m", "L", "unused", num_classes=1))
    rng = np.random.default_rng(42)
    rects = synth_boxes(rng, n, size)
    shapes = {"data": (n, 3, size, size), "coverage-label": (n, 1, 28, 28)}
    for k in LABELS[1:]:
        shapes[k] = (n, 4, 28, 28)
    spec = NetSpec(msg, "TRAIN")
    spec.infer(shapes)
    params = fill_params(spec, seed=1234)
    eng = TrainEngine(NetSpec(msg, "TRAIN"), shapes, params={k: [a.copy() for a in v] for k, v in params.items()}, device=0,
                      solver=SolverParams(base_lr=0.0, momentum=0.9, weight_decay=1e-7, lr_policy="fixed"))      # lr 0: weights stay
    data = {"data": rng.random((n, 3, size, size), dtype=np.float32)}
    eng.host_array("data")[...] = data["data"]
    eng.set_targets(rects, [[0] * len(r) for r in rects], stride=16)      # the five label tensors are generated in HBM
    out = eng.step(seed=5)
    # (1) device-generated targets: bit for bit what the oracle's target generator gives (pinned to the reference's own
    #     function by tests/test_reference_golden.py)
    lab = [D.bounding_box_parameterized_labels(size, size, r, [0] * len(r), 16, 1) for r in rects]
    for j, name in enumerate(LABELS):
        data[name] = np.stack([o[j] for o in lab]).astype(np.float32)
        assert np.array_equal(eng.read_blob(name), data[name]), name
    assert data["coverage-block"].sum() > 100
    # (2) forward: losses and head blobs against the oracle's own forward pass
    ref = RefNet(msg, "TRAIN", {k: [a.copy() for a in v] for k, v in params.items()})
    ref.blobs.update(data)
    ref.dropout_seed = 5
    ref.forward()
    for key in ("loss_bbox", "loss_coverage"):
        assert abs(out[key] - ref.losses[key]) < 1e-3 * abs(ref.losses[key]), key
    assert abs(out["total_loss"] - ref.total_loss()) < 1e-3 * abs(ref.total_loss())
    for name in ("coverage", "bboxes", "pool5/drop_s1", "inception_4a/output", "conv2/norm2"):
        assert rel_err(eng.read_blob(name), ref.blobs[name]) < 1e-3, name
    for name in ("coverage", "bboxes"):      # the head blobs element by element: |a - b| <= 1e-3 |b| + 1e-3 rms(b)
        worst, at = elem_err(eng.read_blob(name), ref.blobs[name])
        assert worst <= 1.0, (name, worst, at)
    # (3) backward of the heads and the top of the net, on the device's own activations
    adopt_device_activations(ref, eng, spec, keep=data)
    grads = ref.backward(stop_at="inception_5a/3x3_reduce")
    for name in ("bboxes", "cvg/classifier", "pool5/drop_s1", "inception_5b/3x3_reduce", "inception_5a/pool"):
        assert rel_err(eng.read_grad(name), ref.diffs[name]) < 1e-3, name
    got = eng.download_grads()
    checked = 0
    for name in ("bbox/regressor", "cvg/classifier", "inception_5b/1x1", "inception_5b/3x3", "inception_5b/5x5_reduce", "inception_5a/pool_proj"):
        for g, r in zip(got[name], grads[name]):
            assert g.shape == r.shape and rel_err(g, r) < 1e-3, name
            assert elem_err(g, r)[0] <= 1.0, (name, elem_err(g, r))
            checked += 1
    assert checked == 12
    eng.close()


def test_full_backward_448_to_conv1_and_an_independent_oracle_step(gpu, capsys):
    """Two things the batch-8 test above leaves open:

    (a) the WHOLE backward at full resolution: batch 2, 448x448, the oracle's backward down to conv1/7x7_s2 on the device's own
        activations - every parameter gradient of the net (59 convolutions) and the stem's blob gradients at 1e-3;
    (b) a fully INDEPENDENT oracle step (its own forward pass, its own ReLU masks and pooling argmaxes): how many mask / argmax
        sites differ from the device's, and what that does to the gradients.  ReLU and MAX pooling are discontinuous, so two
        independently rounded forward passes flip a few near-zero activations / near-tied windows; the bound asserted here is the
        measured effect with a margin (the numbers are printed and quoted in DESIGN.md 2)."""
    from oracle import caffe_ref as R
    n, size = 2, 448
    msg = proto.parse_text(models.googlenet_detectnet_train("m", "L", "unused", num_classes=1))
    rng = np.random.default_rng(7)
    rects = synth_boxes(rng, n, size)
    shapes = {"data": (n, 3, size, size), "coverage-label": (n, 1, 28, 28)}
    for k in LABELS[1:]:
        shapes[k] = (n, 4, 28, 28)
    spec = NetSpec(msg, "TRAIN")
    spec.infer(shapes)
    params = fill_params(spec, seed=1234)
    eng = TrainEngine(NetSpec(msg, "TRAIN"), shapes, params={k: [a.copy() for a in v] for k, v in params.items()}, device=0,
                      solver=SolverParams(base_lr=0.0, momentum=0.9, weight_decay=0.0, lr_policy="fixed"))
    data = {"data": rng.random((n, 3, size, size), dtype=np.float32)}
    eng.host_array("data")[...] = data["data"]
    eng.set_targets(rects, [[0] * len(r) for r in rects], stride=16)
    eng.step(seed=5)
    lab = [D.bounding_box_parameterized_labels(size, size, r, [0] * len(r), 16, 1) for r in rects]
    for j, name in enumerate(LABELS):
        data[name] = np.stack([o[j] for o in lab]).astype(np.float32)
    got = eng.download_grads()

    # (b) first: the independent step
    ref = RefNet(msg, "TRAIN", {k: [a.copy() for a in v] for k, v in params.items()})
    ref.blobs.update(data)
    ref.dropout_seed = 5
    ref.forward()
    ind = ref.backward()
    relu_sites = relu_flips = 0
    for l in spec.layers:
        if l.type == "ReLU":
            a, b = ref.blobs[l.tops[0]] > 0, eng.read_blob(l.tops[0]) > 0
            relu_sites += a.size
            relu_flips += int((a != b).sum())
    pool_sites = pool_flips = 0
    for l in spec.layers:
        if l.type == "Pooling" and str(l.sub("pooling_param").get("pool", "MAX")) == "MAX":
            k, s_, p_ = (int(l.sub("pooling_param").get(q, d)) for q, d in (("kernel_size", 0), ("stride", 1), ("pad", 0)))
            dev_idx = R.max_pool(eng.read_blob(l.bottoms[0]), k, s_, p_, return_index=True)[1]
            pool_sites += dev_idx.size
            pool_flips += int((np.asarray(ref.aux[l.name]) != dev_idx).sum())
    ind_err = {name: max(rel_err(g, r) for g, r in zip(got[name], ind[name])) for name in ind}
    worst = max(ind_err, key=ind_err.get)
    # The decomposition (round 4): the SAME independent oracle pass - its own activations - run backward
    # twice more: (i) with the DEVICE's pooling argmaxes in place of its own, (ii) with the device's argmaxes AND the device's ReLU
    # masks (a site the device holds at zero is zeroed, a site the device holds positive is kept positive: 1e-30 where the oracle
    # had 0 - the values enter the weight gradients, the signs decide the masks).  If the excursion above 1e-3 is those
    # discontinuities and nothing else, (ii) leaves every parameter gradient far inside 1e-3.  One flipped ReLU site moves a
    # whole filter row of the layer above it by (upstream gradient x input window) - against a sum over 1568 pixels that is
    # 1e-3 .. 1e-2 of the blob's largest element, which is why 8 flips in 28.5 M matter as much as 3.5 k argmax flips.
    own_aux, own_blobs = {}, {}
    for l in spec.layers:
        if l.type == "Pooling" and str(l.sub("pooling_param").get("pool", "MAX")) == "MAX":
            k, s_, p_ = (int(l.sub("pooling_param").get(q, d)) for q, d in (("kernel_size", 0), ("stride", 1), ("pad", 0)))
            own_aux[l.name] = ref.aux[l.name]
            ref.aux[l.name] = R.max_pool(eng.read_blob(l.bottoms[0]), k, s_, p_, return_index=True)[1]
    adopted = ref.backward()
    arg_err = {name: max(rel_err(g, r) for g, r in zip(got[name], adopted[name])) for name in adopted}
    arg_worst = max(arg_err, key=arg_err.get)
    for l in spec.layers:
        if l.type == "ReLU":
            t = l.tops[0]
            dev_pos = eng.read_blob(t) > 0
            own_blobs[t] = ref.blobs[t]
            ref.blobs[t] = np.where(dev_pos, np.maximum(ref.blobs[t], np.float32(1e-30)), np.float32(0)).astype(np.float32)
    adopted = ref.backward()
    ref.aux.update(own_aux)
    ref.blobs.update(own_blobs)
    ado_err = {name: max(rel_err(g, r) for g, r in zip(got[name], adopted[name])) for name in adopted}
    ado_worst = max(ado_err, key=ado_err.get)
    with capsys.disabled():
        print("\nindependent oracle step, batch %d at %dx%d: ReLU mask flips %d of %d (%.2e), MAX-pool argmax flips %d of %d (%.2e); "
              "parameter-gradient rel. error: median %.2e, max %.2e (%s); with the device's argmaxes adopted: median %.2e, max %.2e (%s); "
              "with the device's argmaxes AND ReLU masks adopted: median %.2e, max %.2e (%s); %s itself %.2e -> %.2e -> %.2e" % (
                  n, size, size, relu_flips, relu_sites, relu_flips / relu_sites, pool_flips, pool_sites, pool_flips / pool_sites,
                  float(np.median(list(ind_err.values()))), ind_err[worst], worst,
                  float(np.median(list(arg_err.values()))), arg_err[arg_worst], arg_worst,
                  float(np.median(list(ado_err.values()))), ado_err[ado_worst], ado_worst, worst, ind_err[worst], arg_err[worst], ado_err[worst]))
    # measured (MI355X): 8 of 28.5 M ReLU sites (2.8e-7), 3587 of 13.0 M argmaxes (2.8e-4); gradients: median 2.9e-4, max 1.7e-3 independent,
    # max 1.2e-3 with the argmaxes adopted (round 4)
    assert relu_flips <= 1e-5 * relu_sites and pool_flips <= 1e-3 * pool_sites      # a handful of near-zero / near-tied sites ...
    # ... whose effect is the WHOLE excursion: with the device's discontinuous choices (and nothing else of the device's) every
    # parameter gradient of the independent pass is within the north-star tolerance
    assert max(ado_err.values()) < 1e-3, (ado_worst, ado_err[ado_worst])
    assert float(np.median(list(ind_err.values()))) < 1e-3

    # (a) the whole backward on the device's activations: identical masks and argmaxes, what is left is the backward arithmetic
    adopt_device_activations(ref, eng, spec, keep=data)
    grads = ref.backward()
    assert set(grads) == set(got) and len(grads) == 59
    worst_elem = (0.0, None)
    for name in grads:
        for g, r in zip(got[name], grads[name]):
            assert g.shape == r.shape and rel_err(g, r) < 1e-3, name
            e = elem_err(g, r)[0]      # element by element: |a - b| <= 1e-3 |b| + 1e-3 rms(b)
            assert e <= 1.0, (name, e)
            worst_elem = max(worst_elem, (e, name))
    with capsys.disabled():
        print("whole backward on the device's activations: worst element of any of the 118 parameter blobs uses %.3f of its allowance "
              "(1e-3 |b| + 1e-3 rms(b)), in %s" % worst_elem)
    # (blob gradients of the stem; a concatenation's gradient is not compared: the device applies the ReLU masks of the four
    #  producing convolutions to it in place, the oracle keeps the gradient of the concatenated blob)
    for name in ("conv1/7x7_s2", "pool1/3x3_s2", "pool1/norm1", "conv2/3x3_reduce", "conv2/3x3", "conv2/norm2", "pool2/3x3_s2"):
        assert rel_err(eng.read_grad(name), ref.diffs[name]) < 1e-3, name
    eng.close()


def test_config4_batch32_f16_with_fused_decode(gpu):
    n, size, classes = 32, 448, 4
    msg = proto.parse_text(models.googlenet_detectnet_deploy(n, size, size, classes))
    spec = NetSpec(msg, "TEST")
    spec.infer()
    params = fill_params(spec, seed=1234)
    rng = np.random.default_rng(9)
    # raised head biases (as tests/test_gpu_detect.py::test_batched_node_pipeline_matches_oracle): most cells fire and vote
    # for similar rects, so the fused decode / groupRectangles launch has real clusters to find in every class
    params["cvg/classifier"][1][...] = 1.5
    params["bbox/regressor"][0][...] *= 0.05
    params["bbox/regressor"][1][...] = np.tile(np.array([-30, -25, 35, 40], np.float32), classes) + rng.normal(0, 0.5, 4 * classes).astype(np.float32)
    frames = [rng.integers(0, 256, (size, size, 3), dtype=np.uint8) for _ in range(n)]
    maps = {}
    for dtype in ("f32", "f16"):
        eng = Engine(NetSpec(msg, "TEST"), params=params, device=0, dtype=dtype)
        det = FCNObjectDetector(eng, 0.5, 3, 0.2, HeadMapping.detectnet_deploy())
        res = det.run_detector_batch(frames)
        cvg, bb = eng.read_blob("coverage").copy(), eng.read_blob("bboxes").copy()
        maps[dtype] = (cvg, bb)
        total = 0
        candidates = (cvg >= 0.5).sum(axis=(2, 3))      # per (image, class): M of the O(M^2) clustering
        for i in range(n):
            rdet, rlab = D.detect(cvg[i], bb[i], size, size, 16, 0.5, 3, 0.2, fast=True)
            rbox = np.asarray(rdet, dtype=np.int64).reshape(-1, 5)
            if len(rbox):
                rbox = D.resize_detection((size, size), rbox, size, size)
            boxes, labels = res[i]
            assert np.array_equal(boxes, rbox) and np.array_equal(labels, rlab), (dtype, i)      # integer work: bit-exact
            total += len(boxes)
        assert total > 0 and candidates.mean() > 50, (dtype, total, float(candidates.mean()))
        eng.close()
    # f16 storage (activations and weights rounded to halves, f32 accumulation) against the f32 engine: SURVEY 8(d) config 5
    # states the tolerance separately from the f32 path's 1e-3 - half-float rounding is 4.9e-4 per stored value and the
    # stack is 22 convolutions deep; measured 7e-4 / 1.7e-3, held to 5e-3 of the blob's range
    for k, name in enumerate(("coverage", "bboxes")):
        assert rel_err(maps["f16"][k], maps["f32"][k]) < 5e-3, name
