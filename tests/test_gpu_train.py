"""End-to-end parity of the training step (forward + backward + solver) of the DetectNet GoogLeNet training net
(reference: models/train_val.prototxt with the Python data layer's tops) against the CPU oracle (-m gpu)."""
import numpy as np
import pytest

from conftest import rel_err
from gpu_util import adopt_device_activations
from fcn_object_detector_amd import models, proto
from fcn_object_detector_amd.netspec import NetSpec, fill_params
from fcn_object_detector_amd.train import SolverParams, TrainEngine
from oracle import detect_ref as D
from oracle.net_ref import RefNet, RefSolver

pytestmark = pytest.mark.gpu


def make_batch(rng, n, h, w, stride=16):
    data = {"data": rng.random((n, 3, h, w), dtype=np.float32)}
    outs = []
    for i in range(n):
        rects = []
        for _ in range(int(rng.integers(1, 4))):
            bw, bh = int(rng.integers(24, w // 2)), int(rng.integers(24, h // 2))
            rects.append((int(rng.integers(0, w - bw)), int(rng.integers(0, h - bh)), bw, bh))
        outs.append(D.bounding_box_parameterized_labels(h, w, rects, [0] * len(rects), stride, 1))
    for j, name in enumerate(("coverage-label", "bbox-label", "size-block", "obj-block", "coverage-block")):
        data[name] = np.stack([o[j] for o in outs]).astype(np.float32)
    return data


def build(kind="SGD", n=2, h=96, w=128, lr=1e-4):
    msg = proto.parse_text(models.googlenet_detectnet_train("m", "L", "unused", num_classes=1))
    rng = np.random.default_rng(42)
    data = make_batch(rng, n, h, w)
    shapes = {k: v.shape for k, v in data.items()}
    spec = NetSpec(msg, "TRAIN")
    spec.infer(shapes)
    params = fill_params(spec, seed=1234)
    sp = SolverParams(base_lr=lr, momentum=0.9, weight_decay=1e-7, lr_policy="fixed", solver_type=kind)
    eng = TrainEngine(NetSpec(msg, "TRAIN"), shapes, params={k: [a.copy() for a in v] for k, v in params.items()}, device=0, solver=sp)
    ref = RefNet(msg, "TRAIN", {k: [a.copy() for a in v] for k, v in params.items()})
    smsg = proto.parse_text('base_lr: %g momentum: 0.9 weight_decay: 1e-7 lr_policy: "fixed" %s' % (lr, "solver_type: ADAM" if kind == "ADAM" else ""))
    lrm = {l.name: l.lr_mult for l in spec.param_layers()}
    dcm = {l.name: l.decay_mult for l in spec.param_layers()}
    return msg, spec, data, eng, ref, RefSolver(ref, smsg, lrm, dcm), rng


def test_forward_backward_gradients_match_oracle(gpu):
    """Forward losses vs the oracle's own forward; backward vs the oracle's backward evaluated ON THE DEVICE'S ACTIVATIONS.
    (ReLU masks and max-pool argmaxes are discontinuous: with independently rounded forward passes a handful of
    near-zero activations / near-tied windows flip and the flipped gradient spreads over all channels below, which says
    nothing about the backward kernels.  Feeding the oracle the device's activations makes masks and argmaxes identical,
    so what is left is the backward arithmetic itself.)"""
    from oracle import caffe_ref as R
    msg, spec, data, eng, ref, rsolver, rng = build(lr=0.0)          # lr 0: the step leaves the weights alone
    for k, v in data.items():
        eng.host_array(k)[...] = v
    out = eng.step(seed=7)
    ref.blobs.update(data)
    ref.dropout_seed = 7
    ref.forward()
    assert abs(out["loss_bbox"] - ref.losses["loss_bbox"]) < 1e-3 * abs(ref.losses["loss_bbox"])
    assert abs(out["loss_coverage"] - ref.losses["loss_coverage"]) < 1e-3 * abs(ref.losses["loss_coverage"])
    assert abs(out["loss"] - ref.total_loss()) < 1e-3 * abs(ref.total_loss())
    for name in ("coverage", "bboxes", "pool5/drop_s1", "inception_3a/output"):
        assert rel_err(eng.read_blob(name), ref.blobs[name]) < 1e-3, name
    adopt_device_activations(ref, eng, spec, keep=data)      # oracle backward on the device's activations
    grads = ref.backward()
    for name in ("bboxes", "cvg/classifier", "pool5/drop_s1", "inception_5a/1x1", "inception_4e/pool", "pool3/3x3_s2", "inception_3a/5x5",
                 "conv2/norm2", "conv2/3x3", "pool1/norm1", "pool1/3x3_s2", "conv1/7x7_s2"):
        assert rel_err(eng.read_grad(name), ref.diffs[name]) < 1e-4, name
    # a Concat output shares its gradient buffer with the member convolutions, whose in-place ReLU backward has masked
    # their slices by the time the step is over: compare with the members' (masked) gradients
    for mod in ("inception_5b", "inception_4a", "inception_3a"):
        members = [mod + "/1x1", mod + "/3x3", mod + "/5x5", mod + "/pool_proj"]
        assert rel_err(eng.read_grad(mod + "/output"), np.concatenate([ref.diffs[m] for m in members], axis=1)) < 1e-4, mod
    got = eng.download_grads()
    for name, gs in grads.items():
        for g, r in zip(got[name], gs):
            assert g.shape == r.shape
            assert rel_err(g, r) < 2e-4, name
    eng.close()


@pytest.mark.parametrize("kind", ["SGD", "ADAM"])
def test_three_solver_steps_match_oracle(gpu, kind):
    """BASELINE config 3 check at reduced size: loss trajectory and weights after 3 steps within 1e-3 relative."""
    msg, spec, data, eng, ref, rsolver, rng = build(kind=kind, lr=1e-3 if kind == "SGD" else 1e-4)
    losses, rlosses = [], []
    for it in range(3):
        batch = make_batch(np.random.default_rng(42 + it), 2, 96, 128)
        for k, v in batch.items():
            eng.host_array(k)[...] = v
        losses.append(eng.step(seed=100 + it)["loss"])
        ref.blobs.update(batch)
        ref.dropout_seed = 100 + it
        ref.forward()
        rlosses.append(ref.total_loss())
        # the oracle's backward runs on the device's forward pass of this step (identical ReLU masks and pooling argmaxes, see
        # adopt_device_activations): the trajectory then compares the backward kernels and the solver, at the north-star 1e-3
        adopt_device_activations(ref, eng, spec, keep=batch)
        rsolver.apply(ref.backward())
    for a, b in zip(losses, rlosses):
        assert abs(a - b) < 1e-3 * abs(b), (losses, rlosses)
    got = eng.download_params()
    for name, ps in ref.params.items():
        for g, r in zip(got[name], ps):
            assert rel_err(g, r) < 1e-3, name
    eng.close()


def test_reference_lmdb_fronted_net_equals_python_layer_net(gpu):
    """models/train_val.prototxt as shipped (Data tops `data` / 17-channel `label`, Slice into the five label blobs) gives the
    same losses and weight gradients as the Python-layer form of the same net when the label record holds the same tensors."""
    # both engines without autotuning: the same (heuristic) tile shapes, hence the same rounding, ReLU masks and pool argmaxes
    msg = proto.parse_text(models.googlenet_detectnet_train("m", "L", "unused", num_classes=1))
    data = make_batch(np.random.default_rng(42), 2, 96, 128)
    pshapes = {k: v.shape for k, v in data.items()}
    spec = NetSpec(msg, "TRAIN")
    spec.infer(pshapes)
    sp = SolverParams(base_lr=0.0, momentum=0.9, weight_decay=1e-7, lr_policy="fixed")
    eng = TrainEngine(NetSpec(msg, "TRAIN"), pshapes, params=fill_params(spec, seed=1234), device=0, solver=sp, autotune=False)
    for k, v in data.items():
        eng.host_array(k)[...] = v
    want = eng.step(seed=3)
    g_want = eng.download_grads()
    eng.close()
    lmsg = proto.parse_text(models.googlenet_detectnet_train_lmdb(batch=2, num_classes=1))
    record = np.concatenate([data[k] for k in ("coverage-label", "bbox-label", "size-block", "obj-block", "coverage-block")], axis=1)
    assert record.shape[1] == 17
    shapes = {"data": data["data"].shape, "label": record.shape}
    lspec = NetSpec(lmsg, "TRAIN")
    lspec.infer(shapes)
    leng = TrainEngine(NetSpec(lmsg, "TRAIN"), shapes, params=fill_params(lspec, seed=1234), device=0, solver=sp, autotune=False)
    leng.host_array("data")[...] = data["data"]
    leng.host_array("label")[...] = record
    got = leng.step(seed=3)
    for k in ("loss_bbox", "loss_coverage", "total_loss"):
        assert abs(got[k] - want[k]) <= 1e-6 * abs(want[k]), k
    g_got = leng.download_grads()
    for name in ("conv1/7x7_s2", "inception_4a/1x1", "bbox/regressor"):
        assert rel_err(g_got[name][0], g_want[name][0]) < 1e-5, name
    leng.close()


def test_two_stream_step_is_deterministic(gpu, tmp_path):
    """The weight gradients run on a second stream beside the data-gradient chain: 40 short runs from identical
    state and data give identical losses, within rounding of the single-stream step.  (This is the check that found the
    convolution kernel's un-waited tail prefetch overwriting accumulator copies when a weight-gradient kernel shares its CU.)"""
    import os
    import random
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fcn_object_detector_amd", "python"))
    from fcn_object_detector_amd.solver import Solver
    net = tmp_path / "t.prototxt"
    net.write_text(models.googlenet_detectnet_train("data_argumentation_layer", "DataArgumentationLayer", "128,96,16,2,2,synthetic:2,detectnet",
                                                    num_classes=2))
    sol = tmp_path / "s.prototxt"
    sol.write_text('net: "%s"\nbase_lr: 1e-4\nmomentum: 0.9\nweight_decay: 1e-6\nlr_policy: "fixed"\ndisplay: 0\nmax_iter: 100\nsnapshot: 0\n' % net)

    def run():
        s = Solver(str(sol), device=0, log=None, autotune=False)
        random.seed(5)
        s.py_layers[0][1]._color_rng = np.random.default_rng(1234)
        out = [s.step(1)["loss"] for _ in range(3)]
        s.close()
        return out

    old = os.environ.get("FCN_WGRAD_STREAM")
    try:
        os.environ["FCN_WGRAD_STREAM"] = "0"
        single = run()
        os.environ["FCN_WGRAD_STREAM"] = "1"
        ref = run()
        assert np.allclose(ref, single, rtol=1e-5)      # (the two modes may pick different tile shapes: last-bit differences only)
        for i in range(40):
            assert run() == ref, i
    finally:
        if old is None:
            os.environ.pop("FCN_WGRAD_STREAM", None)
        else:
            os.environ["FCN_WGRAD_STREAM"] = old
