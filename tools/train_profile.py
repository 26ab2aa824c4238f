#!/usr/bin/env python
"""GPU: per-launch timing of one training step (forward ops, backward ops, update) of the DetectNet GoogLeNet net."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_boxes  # noqa: E402
from fcn_object_detector_amd import lib as L, models, proto  # noqa: E402
from fcn_object_detector_amd.netspec import NetSpec, fill_params  # noqa: E402
from fcn_object_detector_amd.train import SolverParams, TrainEngine  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    msg = proto.parse_text(models.googlenet_detectnet_train("synthetic", "Boxes", "448,448,16,1,%d,none" % n, num_classes=1))
    shapes = {"data": (n, 3, 448, 448), "coverage-label": (n, 1, 28, 28)}
    for k in ("bbox-label", "size-block", "obj-block", "coverage-block"):
        shapes[k] = (n, 4, 28, 28)
    spec = NetSpec(msg, "TRAIN")
    spec.infer(shapes)
    eng = TrainEngine(NetSpec(msg, "TRAIN"), shapes, params=fill_params(spec, seed=1234), device=0,
                      solver=SolverParams(base_lr=1e-4, momentum=0.9, weight_decay=1e-7, lr_policy="fixed"))
    eng.host_array("data")[...] = np.random.default_rng(0).random((n, 3, 448, 448), dtype=np.float32)
    eng.upload_inputs()
    for it in range(2):
        eng.set_targets(*synth_boxes(np.random.default_rng(it), n), stride=16)
        print("loss", eng.step(seed=it, upload=False)["total_loss"])
    tot = {}
    for label, ops in (("fwd", eng.ops), ("bwd", eng.bwd_ops)):
        rows = eng.time_ops(reps=5, ops=ops)
        t = sum(r[2] for r in rows)
        print("== %s: %d launches, %.3f ms" % (label, len(rows), t))
        by_kind = {}
        for kind, name, ms, fl, by in rows:
            by_kind.setdefault(kind, [0, 0.0, 0.0])
            by_kind[kind][0] += 1
            by_kind[kind][1] += ms
            by_kind[kind][2] += fl
        for k, (c, ms, fl) in sorted(by_kind.items(), key=lambda kv: -kv[1][1]):
            print("   %-14s %3d launches %8.3f ms  %6.1f TF/s" % (k, c, ms, fl / ms / 1e9 if ms else 0))
        for kind, name, ms, fl, by in sorted(rows, key=lambda r: -r[2])[:25]:
            print("      %-12s %-110s %8.1f us %6.1f TF/s" % (kind, name[:110], ms * 1e3, fl / ms / 1e9 if ms else 0))
        tot[label] = t
    print(tot)


if __name__ == "__main__":
    main()
