#!/usr/bin/env python
"""GPU: which stage of the per-frame node path limits DetectorPipeline: stages are switched on one by one."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fcn_object_detector_amd import lib as L, models, proto  # noqa: E402
from fcn_object_detector_amd.detector import DetectorPipeline, HeadMapping  # noqa: E402
from fcn_object_detector_amd.engine import Engine  # noqa: E402
from fcn_object_detector_amd.netspec import NetSpec, fill_params  # noqa: E402

msg = proto.parse_text(models.googlenet_detectnet_deploy(1, 448, 448, 4))
spec = NetSpec(msg, "TEST")
spec.infer()
params = fill_params(spec, seed=1234)
pipe = DetectorPipeline(lambda first: Engine(NetSpec(msg, "TEST"), params=params, device=0, tune_from=first, tune_max_lds_kb=36), depth=4,
                        mapping=HeadMapping.detectnet_deploy())
frame = np.random.default_rng(0).integers(0, 256, (480, 640, 3), dtype=np.uint8)
pipe.run_detector_stream([frame] * 8)
dets = pipe.detectors
n = 400
for stages in ("F", "HF", "HPF", "HPFD", "HPFDR", "PFD", "FD", "FR"):
    for d in dets:
        L.call("fcn_stream_sync", d.engine.stream)
    t0 = time.perf_counter()
    for i in range(n):
        d = dets[i % 4]
        eng = d.engine
        data = eng.blobs["data"]
        if i >= 4 and "R" in stages:
            L.call("fcn_stream_sync", eng.stream)
        if "H" in stages:
            L.call("fcn_memcpy_h2d_async", d._frame_dev.ptr, d._frame_pinned.array.ctypes.data, frame.nbytes, eng.stream)
        if "P" in stages:
            L.call("fcn_preprocess_bgr8", d._frame_dev.ptr, 480, 640, data.ptr, 448, 448, data.cstride, data.upload_shift, d._minmax.ptr, eng.stream)
        eng.forward_enqueue()
        if "D" in stages:
            d.decoder.launch(*d._cvg_args, *d._box_args, eng.stream)
        if "R" in stages:
            d.decoder.fetch_begin(eng.stream)
    for d in dets:
        L.call("fcn_stream_sync", d.engine.stream)
    print("%-6s %.1f frames/s" % (stages, n / (time.perf_counter() - t0)))
import ctypes as C
d = dets[0]
eng = d.engine
e0, e1 = C.c_void_p(), C.c_void_p()
L.call("fcn_event_create", C.byref(e0))
L.call("fcn_event_create", C.byref(e1))
cvg = eng.read_blob("coverage")
print("cells above 0.5 per class:", (cvg[0] >= 0.5).reshape(4, -1).sum(1))
L.call("fcn_event_record", e0, eng.stream)
for _ in range(50):
    d.decoder.launch(*d._cvg_args, *d._box_args, eng.stream)
L.call("fcn_event_record", e1, eng.stream)
L.call("fcn_event_sync", e1)
ms = C.c_float()
L.call("fcn_event_elapsed_ms", e0, e1, C.byref(ms))
print("decode + groupRectangles launch: %.1f us" % (ms.value / 50 * 1e3))
