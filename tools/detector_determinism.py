#!/usr/bin/env python
"""GPU: the whole node path (DetectorPipeline, four frames in flight) on the same frame many times: every result identical?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fcn_object_detector_amd import models, proto  # noqa: E402
from fcn_object_detector_amd.detector import DetectorPipeline, HeadMapping  # noqa: E402
from fcn_object_detector_amd.engine import Engine  # noqa: E402
from fcn_object_detector_amd.netspec import NetSpec, fill_params  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
msg = proto.parse_text(models.googlenet_detectnet_deploy(1, 448, 448, 4))
spec = NetSpec(msg, "TEST")
spec.infer()
params = fill_params(spec, seed=77)
rng = np.random.default_rng(6)
params["cvg/classifier"][1][...] = 1.5                      # most cells fire: the clustering kernel's sliced path
params["bbox/regressor"][0][...] = 0
params["bbox/regressor"][1][...] = np.tile(np.array([-30, -25, 35, 40], np.float32), 4) + rng.normal(0, 0.5, 16).astype(np.float32)
pipe = DetectorPipeline(lambda first: Engine(NetSpec(msg, "TEST"), params=params, device=0, tune_from=first, tune_max_lds_kb=36), depth=4,
                        mapping=HeadMapping.detectnet_deploy())
frames = [rng.integers(0, 256, (480, 640, 3), dtype=np.uint8) for _ in range(3)]
ref = pipe.run_detector_stream(frames)
print("detections per frame:", [len(b) for b, _ in ref])
out = pipe.run_detector_stream(frames[i % 3] for i in range(n))
bad = sum(1 for i, (b, l) in enumerate(out) if not (np.array_equal(b, ref[i % 3][0]) and np.array_equal(l, ref[i % 3][1])))
print("%d of %d frames differ" % (bad, n))
