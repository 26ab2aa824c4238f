#!/usr/bin/env python
"""GPU micro-benchmark of the LRN backward kernel at the two geometries of the training net (and variations)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fcn_object_detector_amd import lib as L  # noqa: E402
from fcn_object_detector_amd.engine import DeviceBuffer  # noqa: E402

L.call("fcn_init", 0)
lib = L.load()
e0, e1 = C.c_void_p(), C.c_void_p()
L.call("fcn_event_create", C.byref(e0))
L.call("fcn_event_create", C.byref(e1))
for (pix, ch) in ((8 * 112 * 112, 64), (8 * 56 * 56, 192), (8 * 56 * 56, 64), (8 * 56 * 56, 256), (8 * 112 * 112, 48), (8 * 112 * 112, 192)):
    n = pix * ch
    bufs = [DeviceBuffer(n * 4, zero=True) for _ in range(5)]
    ones = np.ones(n, np.float32)
    for b in bufs[:4]:
        L.call("fcn_memcpy_h2d_async", b.ptr, ones.ctypes.data, n * 4, None)
    L.call("fcn_device_sync")
    x, y, sc, dy, dx = bufs
    for acc in (0, 1):
        for _ in range(3):
            L.check(lib.fcn_lrn_bwd_f32(x.ptr, y.ptr, sc.ptr, dy.ptr, dx.ptr, pix, ch, ch, ch, 5, 1e-4, 0.75, acc, None))
        L.call("fcn_event_record", e0, None)
        for _ in range(20):
            L.check(lib.fcn_lrn_bwd_f32(x.ptr, y.ptr, sc.ptr, dy.ptr, dx.ptr, pix, ch, ch, ch, 5, 1e-4, 0.75, acc, None))
        L.call("fcn_event_record", e1, None)
        L.call("fcn_event_sync", e1)
        ms = C.c_float()
        L.call("fcn_event_elapsed_ms", e0, e1, C.byref(ms))
        us = ms.value / 20 * 1e3
        print("pixels %7d C %3d acc %d: %7.1f us  %5.2f TB/s (5 tensors of %.1f MB)" % (pix, ch, acc, us, (5 + acc) * n * 4 / us / 1e6, n * 4 / 1e6))
    for b in bufs:
        b.free()
