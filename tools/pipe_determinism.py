#!/usr/bin/env python
"""GPU: do frames that share the GPU (ForwardPipeline) always produce the lone engine's bits?  The same frame goes through
the pipeline many times; every output is compared with the first one."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fcn_object_detector_amd import models, proto  # noqa: E402
from fcn_object_detector_amd.engine import ForwardPipeline  # noqa: E402
from fcn_object_detector_amd.netspec import NetSpec, fill_params  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
h, w = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "448x448").split("x"))
msg = proto.parse_text(models.googlenet_detectnet_deploy(1, h, w, 4))
spec = NetSpec(msg, "TEST")
spec.infer()
params = fill_params(spec, seed=1234)
pipe = ForwardPipeline(lambda: NetSpec(msg, "TEST"), params=params, device=0, depth=4)
x = np.random.default_rng(0).random((1, 3, h, w), dtype=np.float32)
ref = pipe.map([{"data": x}])[0]
bad = 0
outs = pipe.map([{"data": x}] * n)
for i, o in enumerate(outs):
    if not all(np.array_equal(o[k], ref[k]) for k in ref):
        bad += 1
        if bad <= 5:
            d = np.abs(o["bboxes"] - ref["bboxes"])
            print("frame %d differs: %d bbox values, max %.3g" % (i, int((d > 0).sum()), d.max()))
print("%d of %d frames differ from the first result (%dx%d, 4 in flight)" % (bad, n, h, w))
