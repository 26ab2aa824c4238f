#!/usr/bin/env python
"""GPU: per-launch HIP-event times of one forward of the deploy net.  usage: python tools/fwd_ops.py batch f32|f16"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fcn_object_detector_amd import models, proto  # noqa: E402
from fcn_object_detector_amd.engine import Engine  # noqa: E402
from fcn_object_detector_amd.netspec import NetSpec, fill_params  # noqa: E402


def main():
    n, dtype = int(sys.argv[1]), sys.argv[2]
    msg = proto.parse_text(models.googlenet_detectnet_deploy(batch=n))
    spec = NetSpec(msg, "TEST")
    spec.infer()
    eng = Engine(NetSpec(msg, "TEST"), params=fill_params(spec, seed=1234), device=0, dtype=dtype)
    eng.host_array("data")[...] = np.random.default_rng(0).random((n, 3, 448, 448), dtype=np.float32)
    eng.upload_inputs()
    eng.forward_resident(2, use_graph=False)
    rows = eng.time_ops(reps=10)
    tot = sum(r[2] for r in rows)
    by_kind = {}
    for kind, name, ms, fl, by in rows:
        print("%-10s %-64s %8.1f us %7.1f TF/s %7.1f GB/s" % (kind, name[:64], ms * 1e3, fl / ms / 1e9 if ms else 0, by / ms / 1e6 if ms else 0))
        by_kind[kind] = by_kind.get(kind, 0.0) + ms
    print("total %.3f ms  " % tot + "  ".join("%s %.3f" % kv for kv in sorted(by_kind.items(), key=lambda kv: -kv[1])))
    eng.close()


if __name__ == "__main__":
    main()
