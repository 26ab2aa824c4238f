#!/usr/bin/env python
"""GPU: where a batch of 32 frames spends its time in the half-float detector (BASELINE configs[4]): HIP events around the pre-processing,
the forward and the decode + groupRectangles launch, and the wall clock of submit + collect (host side included)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fcn_object_detector_amd import lib as L, models, proto  # noqa: E402
from fcn_object_detector_amd.detector import FCNObjectDetector, HeadMapping  # noqa: E402
from fcn_object_detector_amd.engine import DeviceBuffer, Engine  # noqa: E402
from fcn_object_detector_amd.netspec import NetSpec, fill_params  # noqa: E402

n = 32
msg = proto.parse_text(models.googlenet_detectnet_deploy(n, 448, 448, 4))
spec = NetSpec(msg, "TEST")
spec.infer()
params = fill_params(spec, seed=1234)
brng = np.random.default_rng(9)
params["cvg/classifier"][1][...] = 1.5
params["bbox/regressor"][0][...] *= 0.05
params["bbox/regressor"][1][...] = np.tile(np.array([-30, -25, 35, 40], np.float32), 4) + brng.normal(0, 0.5, 16).astype(np.float32)
eng = Engine(NetSpec(msg, "TEST"), params=params, device=0, dtype="f16")
det = FCNObjectDetector(eng, mapping=HeadMapping.detectnet_deploy())
frames = np.random.default_rng(9).integers(0, 256, (n, 448, 448, 3), dtype=np.uint8)
dev = DeviceBuffer(frames.nbytes, zero=False)
L.call("fcn_memcpy_h2d_async", dev.ptr, frames.ctypes.data, frames.nbytes, eng.stream)
L.call("fcn_device_sync")
layout = [(i * 448 * 448 * 3, 448, 448) for i in range(n)]
det._minmax_batch_holder[:] = [DeviceBuffer(32 * n)]
ev = [C.c_void_p() for _ in range(5)]
for e in ev:
    L.call("fcn_event_create", C.byref(e))
data = eng.blobs["data"]


def run(timed):
    st = eng.stream
    if timed:
        L.call("fcn_event_record", ev[0], st)
    L.call("fcn_preprocess_bgr8_batch", dev.ptr, n, 448, 448, data.ptr, det._half_flag(data), 448, 448, data.cstride, data.upload_shift,
           det._minmax_batch_holder[0].ptr, st)
    if timed:
        L.call("fcn_event_record", ev[1], st)
    eng.forward_enqueue()
    if timed:
        L.call("fcn_event_record", ev[2], st)
    det.decoder.launch(*det._cvg_args, *det._box_args, st)
    if timed:
        L.call("fcn_event_record", ev[3], st)
    det.decoder.fetch_begin(st)
    if timed:
        L.call("fcn_event_record", ev[4], st)
    return det.decoder.fetch(st, begun=True)


for _ in range(3):
    run(False)
res = run(True)
ms = C.c_float()
names = ["pre-processing", "forward", "decode + groupRectangles", "read-back"]
for i, nm in enumerate(names):
    L.call("fcn_event_elapsed_ms", ev[i], ev[i + 1], C.byref(ms))
    print("%-26s %8.1f us" % (nm, ms.value * 1e3))
t0 = time.perf_counter()
for _ in range(10):
    res = run(False)
print("wall clock per batch (enqueue + wait + unpack on the host): %.3f ms; detections %d" % ((time.perf_counter() - t0) / 10 * 1e3, sum(len(r[0]) for r in res)))
