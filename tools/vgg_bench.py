#!/usr/bin/env python
"""GPU: the reference's secondary VGG16 nets — forward of the train/fcn_bbox deploy form at 448x448 and one training step of
train/fcn_bbox at its native configuration (288x288, stride 8, 11 classes; batch 24 in the reference's param_str)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_boxes  # noqa: E402
from fcn_object_detector_amd import lib as L, models, proto  # noqa: E402
from fcn_object_detector_amd.engine import Engine  # noqa: E402
from fcn_object_detector_amd.netspec import NetSpec, fill_params  # noqa: E402
from fcn_object_detector_amd.train import SolverParams, TrainEngine  # noqa: E402


def conv_flops(spec, shapes):
    f = 0.0
    for l in spec.layers:
        if l.type == "Convolution":
            n, co, oh, ow = shapes[l.tops[0]]
            ci = shapes[l.bottoms[0]][1]
            k = int(l.sub("convolution_param").get("kernel_size"))
            f += 2.0 * n * co * oh * ow * ci * k * k
    return f


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    L.call("fcn_init", 0)
    msg = proto.parse_text(models.vgg16_fcn_bbox_deploy(1, 448, 448, 11))
    spec = NetSpec(msg, "TEST")
    shapes = spec.infer()
    eng = Engine(NetSpec(msg, "TEST"), params=fill_params(spec, seed=1), device=0)
    eng.host_array("data")[...] = np.random.default_rng(0).random((1, 3, 448, 448), dtype=np.float32)
    eng.upload_inputs()
    eng.forward_resident(3)
    ms = eng.forward_resident(20) / 20
    fl = conv_flops(spec, shapes)
    print("fcn_bbox deploy 448x448 b1: %.3f ms/frame, %.1f frames/s, %.2f GFLOP/frame, %.1f TF/s" % (ms, 1e3 / ms, fl / 1e9, fl / ms / 1e9))
    rows = eng.time_ops(5)
    for kind, name, t, f, b in sorted(rows, key=lambda r: -r[2])[:8]:
        print("   %-10s %-50s %8.1f us %6.1f TF/s" % (kind, name[:50], t * 1e3, f / t / 1e9 if t else 0))
    eng.close()

    n, size, classes = batch, 288, 11
    msg = proto.parse_text(models.vgg16_fcn_bbox_train("synthetic", "Boxes", "288,288,8,11,%d,none" % n, num_classes=classes))
    shapes = {"data": (n, 3, size, size), "label": (n, 1, size, size)}
    for k in ("bbox-label", "size-block", "obj-block", "coverage-block"):
        shapes[k] = (n, 4 * classes, size // 8, size // 8)
    spec = NetSpec(msg, "TRAIN")
    full = spec.infer(shapes)
    te = TrainEngine(NetSpec(msg, "TRAIN"), shapes, params=fill_params(spec, seed=2), device=0,
                     solver=SolverParams(base_lr=1e-10, momentum=0.9, weight_decay=1e-7))
    rng = np.random.default_rng(1)
    te.host_array("data")[...] = rng.random((n, 3, size, size), dtype=np.float32)
    te.host_array("label")[...] = rng.integers(0, classes, (n, 1, size, size)).astype(np.float32)
    te.upload_inputs()
    tops = ("label", "bbox-label", "size-block", "obj-block", "coverage-block")
    fl = conv_flops(spec, full)
    for it in range(3):
        out = te.step(seed=it, upload=False)
    L.call("fcn_device_sync")
    t0 = time.perf_counter()
    steps = 5
    for it in range(steps):
        out = te.step(seed=10 + it, upload=False)
    L.call("fcn_device_sync")
    dt = (time.perf_counter() - t0) / steps
    print("fcn_bbox train 288x288 b%d: %.2f ms/step, %.1f imgs/s, fwd conv %.1f GFLOP/step -> ~%.1f TF/s (3x fwd), losses %s" % (
        n, dt * 1e3, n / dt, fl / 1e9, 3 * fl / dt / 1e12, {k: round(v, 4) for k, v in out.items()}))
    for label, ops in (("fwd", te.ops), ("bwd", te.bwd_ops)):
        rows = te.time_ops(reps=3, ops=ops)
        by = {}
        for kind, name, ms, f, b in rows:
            by.setdefault(kind, [0, 0.0, 0.0])
            by[kind][0] += 1
            by[kind][1] += ms
            by[kind][2] += f
        print("  %s %.2f ms: " % (label, sum(r[2] for r in rows)) + ", ".join("%s x%d %.2f ms %.0f TF/s" % (k, c, ms, f / ms / 1e9 if ms else 0)
                                                                             for k, (c, ms, f) in sorted(by.items(), key=lambda kv: -kv[1][1])))
    te.close()


if __name__ == "__main__":
    main()
