#!/usr/bin/env python
"""GPU experiment: aggregate batch-1 throughput with K independent engines (own stream, own activations) whose forward graphs
are launched round-robin from one host thread: do kernels of different frames fill each other's gaps?"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fcn_object_detector_amd import lib as L, models, proto  # noqa: E402
from fcn_object_detector_amd.engine import Engine  # noqa: E402
from fcn_object_detector_amd.netspec import NetSpec, fill_params  # noqa: E402

msg = proto.parse_text(models.googlenet_detectnet_deploy(batch=1))
spec = NetSpec(msg, "TEST")
spec.infer()
params = fill_params(spec, seed=1234)
lib = L.load()
import os
for k in [int(v) for v in os.environ.get('DEPTHS', '1,2,3,4').split(',')]:
    engs = [Engine(NetSpec(msg, "TEST"), params=params, device=0) for _ in range(k)]
    for e in engs:
        e.host_array("data")[...] = np.random.default_rng(0).random((1, 3, 448, 448), dtype=np.float32)
        e.forward()
        e.forward_resident(5)
    iters = 300
    for e in engs:
        L.call("fcn_stream_sync", e.stream)
    t0 = time.perf_counter()
    for _ in range(iters):
        for e in engs:
            L.check(lib.fcn_graph_launch(e.graph_core, e.stream))
    for e in engs:
        L.call("fcn_stream_sync", e.stream)
    dt = time.perf_counter() - t0
    print("%d engines in flight: %.1f frames/s (%.3f ms per frame)" % (k, iters * k / dt, dt / (iters * k) * 1e3))
    for e in engs:
        e.close()
