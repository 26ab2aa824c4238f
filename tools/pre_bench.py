#!/usr/bin/env python
"""GPU: HIP-event time of the batched pre-processing (fcn_preprocess_bgr8_batch) of N camera frames into the net's input blob.
usage: python tools/pre_bench.py [n] [h] [w]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fcn_object_detector_amd import lib as L  # noqa: E402
from fcn_object_detector_amd.engine import DeviceBuffer  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
h = int(sys.argv[2]) if len(sys.argv) > 2 else 448
w = int(sys.argv[3]) if len(sys.argv) > 3 else 448
lib = L.load()
L.call("fcn_init", 0)
frames = np.random.default_rng(0).integers(0, 256, (n, h, w, 3), dtype=np.uint8)
src = DeviceBuffer(frames.nbytes, zero=False)
L.call("fcn_memcpy_h2d_async", src.ptr, frames.ctypes.data, frames.nbytes, None)
mm = DeviceBuffer(32 * n)
e0, e1 = C.c_void_p(), C.c_void_p()
L.call("fcn_event_create", C.byref(e0))
L.call("fcn_event_create", C.byref(e1))
for f16, cs in ((3, 8), (1, 8), (0, 4)):
    dst = DeviceBuffer(n * 448 * 448 * cs * (2 if f16 else 4))
    for _ in range(3):
        L.call("fcn_preprocess_bgr8_batch", src.ptr, n, h, w, dst.ptr, f16, 448, 448, cs, -127.0 if not f16 else 0.0, mm.ptr, None)
    L.call("fcn_event_record", e0, None)
    for _ in range(20):
        L.call("fcn_preprocess_bgr8_batch", src.ptr, n, h, w, dst.ptr, f16, 448, 448, cs, -127.0 if not f16 else 0.0, mm.ptr, None)
    L.call("fcn_event_record", e1, None)
    L.call("fcn_event_sync", e1)
    ms = C.c_float()
    L.call("fcn_event_elapsed_ms", e0, e1, C.byref(ms))
    print("%d frames %dx%d -> 448x448 %s: %.1f us per batch" % (n, h, w, ("f16 x 8, whole pixels" if f16 == 3 else "f16 x 8") if f16 else "f32 x 4", ms.value / 20 * 1e3))
