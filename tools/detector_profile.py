#!/usr/bin/env python
"""GPU: host-side cost of the per-frame detector path (DetectorPipeline) under cProfile."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fcn_object_detector_amd import models, proto  # noqa: E402
from fcn_object_detector_amd.detector import DetectorPipeline, HeadMapping  # noqa: E402
from fcn_object_detector_amd.engine import Engine  # noqa: E402
from fcn_object_detector_amd.netspec import NetSpec, fill_params  # noqa: E402

msg = proto.parse_text(models.googlenet_detectnet_deploy(1, 448, 448, 4))
spec = NetSpec(msg, "TEST")
spec.infer()
params = fill_params(spec, seed=1234)
pipe = DetectorPipeline(lambda first: Engine(NetSpec(msg, "TEST"), params=params, device=0, tune_from=first, tune_max_lds_kb=36), depth=4,
                        mapping=HeadMapping.detectnet_deploy())
frames = [np.random.default_rng(i).integers(0, 256, (480, 640, 3), dtype=np.uint8) for i in range(8)]
pipe.run_detector_stream(frames)
n = 400
t0 = time.perf_counter()
pipe.run_detector_stream(frames[i % 8] for i in range(n))
print("%.1f frames/s" % (n / (time.perf_counter() - t0)))
pr = cProfile.Profile()
pr.enable()
pipe.run_detector_stream(frames[i % 8] for i in range(n))
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
