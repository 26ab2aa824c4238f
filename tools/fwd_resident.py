#!/usr/bin/env python
"""GPU: N forwards of the deploy net on inputs resident in HBM, as plain launches (for rocprofv3 --pmc passes: the byte counters
of the whole run divided by N forwards = HBM bytes per forward).  usage: python tools/fwd_resident.py batch f32|f16 N"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fcn_object_detector_amd import models, proto  # noqa: E402
from fcn_object_detector_amd.engine import Engine  # noqa: E402
from fcn_object_detector_amd.netspec import NetSpec, fill_params  # noqa: E402


def main():
    n, dtype, reps = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
    msg = proto.parse_text(models.googlenet_detectnet_deploy(batch=n))
    spec = NetSpec(msg, "TEST")
    spec.infer()
    eng = Engine(NetSpec(msg, "TEST"), params=fill_params(spec, seed=1234), device=0, dtype=dtype)
    eng.host_array("data")[...] = np.random.default_rng(0).random((n, 3, 448, 448), dtype=np.float32)
    eng.upload_inputs()
    eng.forward_resident(2, use_graph=False)
    ms = eng.forward_resident(reps, use_graph=False) / reps
    print(json.dumps({"batch": n, "dtype": dtype, "forwards_timed": reps, "forwards_total": reps + 2, "launches_per_forward": len(eng.ops),
                      "ms_per_forward": round(ms, 4)}))
    eng.close()


if __name__ == "__main__":
    main()
