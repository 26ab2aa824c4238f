#!/usr/bin/env python
"""GPU: repeat a 2-iteration solver run from identical state and data many times and report how often the second loss differs
(a determinism check for the multi-stream training step).  usage: race_hunt.py [reps]"""
import os
import random
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "fcn_object_detector_amd", "python"))
from fcn_object_detector_amd import models  # noqa: E402
from fcn_object_detector_amd.solver import Solver  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
tmp = tempfile.mkdtemp()
net = os.path.join(tmp, "t.prototxt")
open(net, "w").write(models.googlenet_detectnet_train("data_argumentation_layer", "DataArgumentationLayer", "128,96,16,2,2,synthetic:2,detectnet",
                                                     num_classes=2))
sol = os.path.join(tmp, "s.prototxt")
open(sol, "w").write('net: "%s"\nbase_lr: 1e-4\nmomentum: 0.9\nweight_decay: 1e-6\nlr_policy: "fixed"\ndisplay: 0\nmax_iter: 100\nsnapshot: 0\n' % net)


def run(dev=True, steps=3):
    s = Solver(sol, device=0, log=None, autotune=False)
    lay = s.py_layers[0][1]
    lay.device_targets = dev
    random.seed(5)
    lay._color_rng = np.random.default_rng(1234)
    out = [s.step(1)["loss"] for _ in range(steps)]
    s.close()
    return out


for cfg in ({}, {"FCN_WGRAD_STREAM": "0"}):
    os.environ.pop("FCN_WGRAD_STREAM", None)
    os.environ.update(cfg)
    ref = run()
    bad = []
    for i in range(reps):
        got = run(dev=True)
        if got != ref:
            bad.append((i, [g - r for g, r in zip(got, ref)]))
    print("%-30s %d / %d runs differ %s" % (cfg or "two streams (default)", len(bad), reps, bad[:3]))
