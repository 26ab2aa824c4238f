"""The reference's secondary net train/fcn_bbox (VGG16 + FCN-8s score branch + x4 bilinear bbox branch): kernels it adds
(softmax, SoftmaxWithLoss, depthwise deconvolution backward, Eltwise SUM backward) and the net end to end (-m gpu)."""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_err
from fcn_object_detector_amd import lib as L
from fcn_object_detector_amd import models, proto
from fcn_object_detector_amd.engine import DeviceBuffer, Engine
from fcn_object_detector_amd.netspec import NetSpec, fill_params
from fcn_object_detector_amd.train import SolverParams, TrainEngine
from gpu_util import dev_from, dev_to, nchw, nhwc
from oracle import caffe_ref as R
from oracle import detect_ref as D
from oracle.net_ref import RefNet

pytestmark = pytest.mark.gpu


def test_softmax_forward(gpu):
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((2, 11, 5, 7)) * 4).astype(np.float32)
    xd, yd = dev_from(nhwc(x, 12)), dev_from(np.zeros((2, 5, 7, 16), np.float32))
    L.call("fcn_softmax_fwd_f32", xd.ptr, yd.ptr, 70, 11, 12, 16, None)
    assert rel_err(nchw(dev_to(yd, (2, 5, 7, 16)), 11), R.softmax(x, 1)) < 1e-6


@pytest.mark.parametrize("normalize,ignore", [(False, None), (True, None), (True, 3), (False, 0)])
@pytest.mark.parametrize("pixels_hw", [(6, 5), (150, 160)])          # one workgroup / many workgroups + fixed-order reduce
def test_softmax_loss_forward_backward(gpu, normalize, ignore, pixels_hw):
    h, w = pixels_hw
    rng = np.random.default_rng(2)
    n, c = 2, 5
    x = (rng.standard_normal((n, c, h, w)) * 3).astype(np.float32)
    lab = rng.integers(0, c, (n, 1, h, w)).astype(np.float32)
    xd, ld = dev_from(nhwc(x, 8)), dev_from(nhwc(lab, 4))
    dxd = dev_from(np.full((n, h, w, 8), 5.0, np.float32))
    lossd = dev_from(np.zeros(4, np.float32))
    ws = DeviceBuffer(int(L.load().fcn_softmax_loss_workspace_bytes()), zero=True)
    args = (xd.ptr, ld.ptr, dxd.ptr, lossd.ptr, n, n * h * w, c, 8, 4, int(normalize), 0 if ignore is None else 1,
            0 if ignore is None else ignore, 2.0, ws.ptr, None)
    L.call("fcn_softmax_loss_f32", *args)
    want = R.softmax_loss(x, lab, normalize, ignore)
    got = float(dev_to(lossd, (4,))[0])
    assert abs(got - want) < 1e-5 * abs(want)
    g = dev_to(dxd, (n, h, w, 8))
    assert rel_err(nchw(g, c), R.softmax_loss_grad(x, lab, normalize, ignore, 2.0)) < 1e-5
    assert np.all(g[..., c:] == 5.0)                              # padded channels untouched
    L.call("fcn_softmax_loss_f32", *args)                         # bit-reproducible
    assert float(dev_to(lossd, (4,))[0]) == got and np.array_equal(dev_to(dxd, (n, h, w, 8)), g)


@pytest.mark.parametrize("k,s,p,h,w,c", [(4, 2, 1, 5, 6, 11), (8, 4, 2, 3, 4, 44), (16, 8, 4, 4, 3, 3)])
def test_depthwise_deconvolution_backward(gpu, k, s, p, h, w, c):
    rng = np.random.default_rng(3)
    wt = R.bilinear_filler((c, 1, k, k)) * rng.uniform(0.5, 1.5, (c, 1, 1, 1)).astype(np.float32)
    oh, ow = R.deconv_out(h, k, p, s), R.deconv_out(w, k, p, s)
    dy = rng.standard_normal((2, c, oh, ow)).astype(np.float32)
    want = R.deconv2d_backward_data(dy, wt, p, s, c)
    co4 = (c + 3) // 4 * 4
    dyd = dev_from(nhwc(dy, co4 + 4, 4))
    wd = dev_from(wt.reshape(c, k, k))
    base = rng.standard_normal((2, c, h, w)).astype(np.float32)
    dxd = dev_from(nhwc(base, co4))
    L.call("fcn_deconv_depthwise_bwd_f32", dyd.ptr, wd.ptr, dxd.ptr, 2, h, w, c, co4, k, s, p, oh, ow, co4 + 4, 4, 1, None)
    assert rel_err(nchw(dev_to(dxd, (2, h, w, co4)), c), base + want) < 1e-5
    L.call("fcn_deconv_depthwise_bwd_f32", dyd.ptr, wd.ptr, dxd.ptr, 2, h, w, c, co4, k, s, p, oh, ow, co4 + 4, 4, 0, None)
    assert rel_err(nchw(dev_to(dxd, (2, h, w, co4)), c), want) < 1e-5


def _vgg_batch(rng, n, size, classes, stride=8):
    data = {"data": rng.random((n, 3, size, size), dtype=np.float32),
            "label": rng.integers(0, classes, (n, 1, size, size)).astype(np.float32)}
    outs = []
    for _ in range(n):
        rects, labels = [], []
        for _ in range(int(rng.integers(1, 3))):
            bw, bh = int(rng.integers(16, size // 2)), int(rng.integers(16, size // 2))
            rects.append((int(rng.integers(0, size - bw)), int(rng.integers(0, size - bh)), bw, bh))
            labels.append(int(rng.integers(0, classes)))
        outs.append(D.bounding_box_parameterized_labels(size, size, rects, labels, stride, classes))
    for j, name in enumerate(("bbox-label", "size-block", "obj-block", "coverage-block")):
        data[name] = np.stack([o[j + 1] for o in outs]).astype(np.float32)
    return data


def test_fcn_bbox_train_net_matches_oracle(gpu):
    """train/fcn_bbox/train_val.prototxt at reduced size (64x64, 3 classes): losses, activations, blob and weight gradients."""
    classes, n, size = 3, 2, 64
    msg = proto.parse_text(models.vgg16_fcn_bbox_train("m", "L", "unused", num_classes=classes))
    rng = np.random.default_rng(5)
    data = _vgg_batch(rng, n, size, classes)
    shapes = {k: v.shape for k, v in data.items()}
    spec = NetSpec(msg, "TRAIN")
    spec.infer(shapes)
    params = fill_params(spec, seed=99)
    eng = TrainEngine(NetSpec(msg, "TRAIN"), shapes, params={k: [a.copy() for a in v] for k, v in params.items()}, device=0,
                      solver=SolverParams(base_lr=0.0, momentum=0.9), autotune=False)
    ref = RefNet(msg, "TRAIN", {k: [a.copy() for a in v] for k, v in params.items()})
    for k, v in data.items():
        eng.host_array(k)[...] = v
    out = eng.step(seed=3)
    ref.blobs.update(data)
    ref.dropout_seed = 3
    ref.forward()
    for name in ("loss", "loss_bbox"):
        assert abs(out[name] - ref.losses[name]) < 1e-3 * abs(ref.losses[name]), name
    assert abs(out["total_loss"] - ref.total_loss()) < 1e-3 * abs(ref.total_loss())
    for name in ("pool3", "dropout5", "upscore_pool5_bbox", "fuse_pool4", "fuse_pool3", "upscore_pool3"):
        assert rel_err(eng.read_blob(name), ref.blobs[name]) < 1e-3, name
    # backward on the device's activations (same reasoning as tests/test_gpu_train.py)
    for name in list(ref.blobs):
        if name in eng.blobs and len(eng.blobs[name].shape) == 4 and name not in data:
            ref.blobs[name] = eng.read_blob(name).copy()
    for l in spec.layers:
        if l.type == "Pooling":
            ref.aux[l.name] = R.max_pool(ref.blobs[l.bottoms[0]], 2, 2, 0, return_index=True)[1]
    grads = ref.backward()
    for name in ("upscore_pool3", "fuse_pool3", "upscore_pool4", "score_pool3", "fuse_pool4", "score_conv5", "upscore_pool5_bbox",
                 "score_conv5_bbox", "dropout5", "pool5", "pool4", "pool3", "pool1"):
        assert rel_err(eng.read_grad(name), ref.diffs[name]) < 1e-4, name
    got = eng.download_grads()
    for name, gs in grads.items():
        for g, r in zip(got[name], gs):
            assert g.shape == r.shape
            assert rel_err(g, r) < 2e-4 or (not np.any(r) and not np.any(g)), name
    # frozen bilinear deconvolutions stay what the filler made them after a real update
    eng.solver.base_lr = 1e-3
    eng.step(seed=4)
    after = eng.download_params()
    for name in ("upscore_pool5_bbox", "upscore_pool5", "upscore_pool4", "upscore_pool3"):
        assert np.array_equal(after[name][0], params[name][0]), name
    assert not np.array_equal(after["score_pool3"][0], params["score_pool3"][0])
    eng.close()


def test_fcn_bbox_deploy_forward_and_detection_heads(gpu):
    """Inference form: `pool_score` (softmax of fuse_pool3, stride 8) and `upscore_pool5_bbox`, the blobs the node reads."""
    classes = 4
    msg = proto.parse_text(models.vgg16_fcn_bbox_deploy(1, 96, 64, classes))
    spec = NetSpec(msg, "TEST")
    spec.infer()
    params = fill_params(spec, seed=7)
    eng = Engine(NetSpec(msg, "TEST"), params={k: [a.copy() for a in v] for k, v in params.items()}, device=0, autotune=False)
    x = np.random.default_rng(0).random((1, 3, 96, 64), dtype=np.float32)
    eng.host_array("data")[...] = x
    out = eng.forward()
    ref = RefNet(msg, "TEST", params)
    ref.blobs["data"] = x
    ref.forward()
    assert eng.shapes["pool_score"] == (1, classes, 12, 8) and eng.shapes["upscore_pool5_bbox"] == (1, 4 * classes, 12, 8)
    for name in ("pool_score", "upscore_pool5_bbox", "upscore_pool3"):
        assert rel_err(eng.read_blob(name), ref.blobs[name]) < 1e-3, name
    assert np.allclose(out["pool_score"].sum(axis=1), 1.0, atol=1e-5)
    eng.close()


def test_bounding_box_train_net_frozen_layers_and_adam(gpu):
    """train/bounding_box/train_val.prototxt at reduced size: conv1_1..conv3_3 frozen (lr_mult 0) get no weight gradient and
    the backward pass stops above them (Caffe's propagate_down); the rest matches the oracle; Adam moves only what learns."""
    classes, n, size = 2, 2, 64
    msg = proto.parse_text(models.vgg16_bounding_box_train("m", "L", "unused", num_classes=classes))
    rng = np.random.default_rng(6)
    data = _vgg_batch(rng, n, size, classes)
    data["coverage-label"] = (data["coverage-block"][:, ::4] > 0).astype(np.float32)
    del data["label"]
    shapes = {k: v.shape for k, v in data.items()}
    spec = NetSpec(msg, "TRAIN")
    spec.infer(shapes)
    params = fill_params(spec, seed=5)
    eng = TrainEngine(NetSpec(msg, "TRAIN"), shapes, params={k: [a.copy() for a in v] for k, v in params.items()}, device=0,
                      solver=SolverParams(base_lr=1e-4, momentum=0.9, momentum2=0.999, solver_type="ADAM", lr_policy="step", gamma=0.1,
                                          stepsize=10000), autotune=False)
    kinds = [(op.kind, op.name.split(" ")[0]) for op in eng.bwd_ops]
    wg = {nm for k, nm in kinds if k == "wgrad"}
    assert "conv4_1" in wg and "cvg/classifier" in wg and not any(nm.startswith(("conv1_", "conv2_", "conv3_")) for nm in wg)
    assert not any(k == "dgrad" and nm.startswith(("conv4_1", "conv3_", "conv2_", "conv1_")) for k, nm in kinds)
    assert "pool3" not in eng.grad_blobs and "conv4_1" in eng.grad_blobs
    ref = RefNet(msg, "TRAIN", {k: [a.copy() for a in v] for k, v in params.items()})
    for k, v in data.items():
        eng.host_array(k)[...] = v
    out = eng.step(seed=9)
    ref.blobs.update(data)
    ref.dropout_seed = 9
    ref.forward()
    assert abs(out["total_loss"] - ref.total_loss()) < 1e-3 * abs(ref.total_loss())
    for name in ("conv5_3/upsample", "coverage", "bboxes"):
        assert rel_err(eng.read_blob(name), ref.blobs[name]) < 1e-3, name
    for name in list(ref.blobs):
        if name in eng.blobs and len(eng.blobs[name].shape) == 4 and name not in data:
            ref.blobs[name] = eng.read_blob(name).copy()
    for l in spec.layers:
        if l.type == "Pooling":
            ref.aux[l.name] = R.max_pool(ref.blobs[l.bottoms[0]], 2, 2, 0, return_index=True)[1]
    grads = ref.backward()
    got = eng.download_grads()
    for name in ("conv4_1", "conv5_3", "cvg/classifier", "bbox/regressor"):
        for g, r in zip(got[name], grads[name]):
            assert rel_err(g, r) < 2e-4, name
    assert not np.any(got["conv3_3"][0]) and not np.any(got["conv1_1"][0])
    after = eng.download_params()
    for name in ("conv1_1", "conv3_3", "conv5_3/upsample"):
        assert np.array_equal(after[name][0], params[name][0]), name
    assert not np.array_equal(after["conv4_1"][0], params["conv4_1"][0])
    eng.close()


def test_bounding_box_deploy_pyramid_pooling_forward(gpu):
    """train/bounding_box/deploy.prototxt at its native 448x448 (the pyramid pools are sized for a 56x56 conv4_3), batch 1:
    AVE pooling to 1x1..7x7, 1x1 convolutions on 1-49 pixels, bilinear deconvolutions k56/s28 .. k8/s4, a copied Concat
    (pool4 also feeds conv5_1), dropout as identity, sigmoid head."""
    msg = proto.parse_text(models.vgg16_bounding_box_deploy(1, 448, 448, 3))
    spec = NetSpec(msg, "TEST")
    spec.infer()
    params = fill_params(spec, seed=3)
    eng = Engine(NetSpec(msg, "TEST"), params={k: [a.copy() for a in v] for k, v in params.items()}, device=0, autotune=False)
    x = np.random.default_rng(2).random((1, 3, 448, 448), dtype=np.float32)
    eng.host_array("data")[...] = x
    out = eng.forward()
    ref = RefNet(msg, "TEST", params)
    ref.blobs["data"] = x
    rb = ref.forward()
    assert eng.shapes["conv4_3/conv5_3/concat"] == (1, 1536, 28, 28)
    for name in ("pool4/1x1", "conv4_3/2x2", "conv4_3/4x4/upsample", "conv4_3/1x1/upsample", "conv5_3", "conv4_3/conv5_3/concat"):
        assert rel_err(eng.read_blob(name), rb[name]) < 1e-3, name
    for name in ("coverage", "bboxes"):
        assert rel_err(out[name], rb[name]) < 1e-3, name
    eng.close()
