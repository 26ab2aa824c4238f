#!/usr/bin/env python
"""Regenerates the golden fixtures in tests/golden/ from the CPU oracle (oracle/).

The reference ships no golden vectors, and its engine (Caffe / OpenCV, Python 2) cannot run here, so these
vectors are produced by this repo's oracle — whose layer ops are cross-checked against torch CPU ops and whose target
generator reproduces the survey's independent known-answer tests (tests/test_oracle.py).  A fixture is DATA: seeded
inputs + expected outputs.   usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from fcn_object_detector_amd import models, proto  # noqa: E402
from fcn_object_detector_amd.netspec import NetSpec, fill_params  # noqa: E402
from oracle import caffe_ref as R  # noqa: E402
from oracle import detect_ref as D  # noqa: E402
from oracle.net_ref import RefNet  # noqa: E402


def layers():
    rng = np.random.default_rng(2024)
    out = {}
    x = rng.standard_normal((1, 12, 9, 11)).astype(np.float32)
    w3 = (rng.standard_normal((20, 12, 3, 3)) * 0.1).astype(np.float32)
    b3 = rng.standard_normal(20).astype(np.float32)
    out.update(conv_x=x, conv_w=w3, conv_b=b3, conv_y=R.relu(R.conv2d(x, w3, b3, 1, 1)))
    x7 = rng.random((1, 3, 21, 17)).astype(np.float32) - np.float32(127)
    w7 = (rng.standard_normal((8, 3, 7, 7)) * 0.05).astype(np.float32)
    out.update(conv7_x=x7, conv7_w=w7, conv7_y=R.conv2d(x7, w7, None, 3, 2))
    p = rng.standard_normal((2, 8, 15, 14)).astype(np.float32)
    y, idx = R.max_pool(p, 3, 2, 0, return_index=True)
    out.update(pool_x=p, pool_y=y, pool_idx=idx.astype(np.int32), pool31_y=R.max_pool(p, 3, 1, 1))
    l = (rng.standard_normal((1, 16, 5, 6)) * 30).astype(np.float32)
    out.update(lrn_x=l, lrn_y=R.lrn_across(l, 5, 1e-4, 0.75, 1.0))
    d = rng.standard_normal((1, 4, 5, 5)).astype(np.float32)
    out.update(deconv_x=d, deconv_y=R.deconv2d(d, R.bilinear_filler((4, 1, 8, 8)), None, 2, 4, group=4))
    return out


def net():
    msg = proto.parse_text(models.googlenet_detectnet_deploy(1, 64, 96, 2))
    spec = NetSpec(msg, "TEST")
    spec.infer()
    params = fill_params(spec, seed=1234)
    frame = np.random.default_rng(0).integers(0, 256, (64, 96, 3), dtype=np.uint8)
    x = D.preprocess_frame(frame, 96, 64)[None]
    ref = RefNet(msg, "TEST", params)
    ref.blobs["data"] = x
    b = ref.forward()
    return dict(frame=frame, data=x, coverage=b["coverage"], bboxes=b["bboxes"], pool3=b["pool3/3x3_s2"][:, :16],
                inc4c_sum=np.array([float(b["inception_4c/output"].astype(np.float64).sum())]))


def detect():
    rng = np.random.default_rng(77)
    cvg = (rng.random((1, 3, 28, 28)) * 0.1).astype(np.float32)
    bb = (rng.standard_normal((1, 12, 28, 28)) * 2).astype(np.float32)
    for k, (x0, y0, w, h) in enumerate([(60, 80, 100, 90), (250, 200, 120, 140), (300, 40, 70, 64)]):
        for cy in range(y0 // 16, (y0 + h) // 16 + 1):
            for cx in range(x0 // 16, (x0 + w) // 16 + 1):
                cvg[0, k, cy, cx] = 0.55 + 0.4 * rng.random()
                j = np.round(rng.standard_normal(4) * 3) / 2.0            # half-integers exercise round-half-even
                bb[0, 4 * k:4 * k + 4, cy, cx] = [x0 - cx * 16 + j[0], y0 - cy * 16 + j[1], x0 + w - cx * 16 + j[2], y0 + h - cy * 16 + j[3]]
    det, lab = D.detect(cvg[0], bb[0], 448, 448, 16, 0.5, 3, 0.2)
    det_t, lab_t = D.detect(cvg[0], bb[0], 448, 448, 16, 0.5, 3, 0.2, round_mode="trunc")
    return dict(cvg=cvg, bbox=bb, det=det, lab=lab, det_trunc=det_t, lab_trunc=lab_t)


def targets():
    rects = [[(100, 120, 80, 60)], [(40, 64, 120, 200), (300, 310, 64, 48), (20, 20, 60, 70)], []]
    labels = [[0], [1, 2, 1], []]
    outs = [D.bounding_box_parameterized_labels(448, 448, r, l, 16, 3) for r, l in zip(rects, labels)]
    names = ("fg", "bbox", "size", "obj", "cvg")
    d = {n: np.stack([o[i] for o in outs]).astype(np.float32) for i, n in enumerate(names)}
    d["rects"] = np.array([r for rs in rects for r in rs], np.int32)
    d["labels"] = np.array([l for ls in labels for l in ls], np.int32)
    d["offsets"] = np.array([0, 1, 4, 4], np.int32)
    return d


def train_batch():
    """Inputs of the training fixture: a seeded batch for the DetectNet training net at 64x96 (shared with the tests)."""
    rng = np.random.default_rng(31)
    rects = [[(10, 8, 40, 30)], [(30, 20, 50, 36), (4, 4, 24, 28)]]
    lab = [D.bounding_box_parameterized_labels(64, 96, r, [0] * len(r), 16, 1) for r in rects]
    data = {"data": rng.random((2, 3, 64, 96), dtype=np.float32)}
    for j, name in enumerate(("coverage-label", "bbox-label", "size-block", "obj-block", "coverage-block")):
        data[name] = np.stack([o[j] for o in lab]).astype(np.float32)
    return data


def train():
    """One solver iteration (SGD, momentum 0.9, lr 1e-3, weight decay 1e-6, dropout seed 11) of the DetectNet training net:
    losses, a few blob / weight gradients and weights after the update."""
    from oracle.net_ref import RefSolver
    msg = proto.parse_text(models.googlenet_detectnet_train("m", "L", "unused", num_classes=1))
    data = train_batch()
    spec = NetSpec(msg, "TRAIN")
    spec.infer({k: v.shape for k, v in data.items()})
    params = fill_params(spec, seed=4321)
    ref = RefNet(msg, "TRAIN", {k: [a.copy() for a in v] for k, v in params.items()})
    ref.blobs.update(data)
    ref.dropout_seed = 11
    ref.forward()
    grads = ref.backward()
    smsg = proto.parse_text('base_lr: 0.001 momentum: 0.9 weight_decay: 1e-6 lr_policy: "fixed"')
    RefSolver(ref, smsg, {l.name: l.lr_mult for l in spec.param_layers()}, {l.name: l.decay_mult for l in spec.param_layers()}).apply(grads)
    out = dict(loss_bbox=np.array([ref.losses["loss_bbox"]]), loss_coverage=np.array([ref.losses["loss_coverage"]]),
               total=np.array([ref.total_loss()]), coverage=ref.blobs["coverage"], bboxes=ref.blobs["bboxes"],
               d_bboxes=ref.diffs["bboxes"], d_pool5=ref.diffs["pool5/drop_s1"])
    for name in ("conv1/7x7_s2", "inception_4a/1x1", "bbox/regressor"):
        key = name.replace("/", "__")
        out["dw_" + key] = grads[name][0]
        out["db_" + key] = grads[name][1]
        out["w_after_" + key] = ref.params[name][0]
    return out


if __name__ == "__main__":
    for name, fn in (("layers", layers), ("net_64x96", net), ("detect", detect), ("targets", targets), ("train_64x96", train)):
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **fn())
        print(name, os.path.getsize(path) // 1024, "KiB")
