#!/usr/bin/env python
"""GPU: per-launch timing of the deploy net's forward (models/deploy.prototxt geometry) at a given batch / dtype.
usage: python tools/fwd_profile.py [batch] [f32|f16]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fcn_object_detector_amd import models, proto  # noqa: E402
from fcn_object_detector_amd.engine import Engine  # noqa: E402
from fcn_object_detector_amd.netspec import NetSpec, fill_params  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
    msg = proto.parse_text(models.googlenet_detectnet_deploy(batch=n))
    spec = NetSpec(msg, "TEST")
    spec.infer()
    eng = Engine(NetSpec(msg, "TEST"), params=fill_params(spec, seed=1234), device=0, dtype=dtype)
    eng.host_array("data")[...] = np.random.default_rng(0).random((n, 3, 448, 448), dtype=np.float32)
    eng.forward()
    rows = eng.time_ops(reps=20)
    t = sum(r[2] for r in rows)
    fl = sum(r[3] for r in rows)
    print("== %d launches, sum of isolated launches %.3f ms, %.1f TF/s" % (len(rows), t, fl / t / 1e9))
    for kind, name, ms, f, by in rows:
        print("  %-10s %-84s %7.1f us %6.1f TF/s %6.1f GFLOP" % (kind, name[:84], ms * 1e3, f / ms / 1e9 if ms else 0, f / 1e9))


if __name__ == "__main__":
    main()
