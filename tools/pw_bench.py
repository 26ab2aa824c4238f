#!/usr/bin/env python
"""GPU: HIP-event times of the half-float pooling / LRN launches of the deploy net at batch N (default 32), one by one on random
blobs.  usage: python tools/pw_bench.py [batch]   ($FCN_LIB_PATH names an experiment build)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fcn_object_detector_amd import lib as L  # noqa: E402
from fcn_object_detector_amd.engine import DeviceBuffer  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    lib = L.load()
    L.call("fcn_init", 0)
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.call("fcn_event_create", C.byref(e0))
    L.call("fcn_event_create", C.byref(e1))
    rng = np.random.default_rng(0)

    def timed(name, fn, nbytes, reps=20):
        for _ in range(3):
            fn()
        L.call("fcn_event_record", e0, None)
        for _ in range(reps):
            fn()
        L.call("fcn_event_record", e1, None)
        L.call("fcn_event_sync", e1)
        ms = C.c_float()
        L.call("fcn_event_elapsed_ms", e0, e1, C.byref(ms))
        us = ms.value / reps * 1e3
        print("%-34s %8.1f us  %7.2f TB/s" % (name, us, nbytes / us / 1e6))
        return us

    def blob(h, w, c):
        a = (rng.standard_normal(n * h * w * c) * 3).astype(np.float16)
        d = DeviceBuffer(a.nbytes, zero=False)
        L.call("fcn_memcpy_h2d_async", d.ptr, a.ctypes.data, a.nbytes, None)
        L.call("fcn_device_sync")
        return d

    total = 0.0
    # pool1 + norm1 (224 -> 112, 64 channels), norm2 + pool2 (112 -> 56, 192 channels)
    for name, h, c, first in (("pool1/3x3_s2+pool1/norm1", 224, 64, 0), ("conv2/norm2+pool2/3x3_s2", 112, 192, 1)):
        oh = (h - 3 + 1) // 2 + 1
        x, y = blob(h, h, c), DeviceBuffer(n * oh * oh * c * 2, zero=False)
        total += timed(name, lambda: L.check(lib.fcn_maxpool_lrn5_fwd_f16(x.ptr, y.ptr, n, h, h, c, c, 3, 2, 0, oh, oh, c, first, 1e-4, 0.75, 1.0, None)),
                       n * (h * h + oh * oh) * c * 2)
        x.free(); y.free()
    # the stand-alone LRN on norm2's blob
    x, y = blob(112, 112, 192), DeviceBuffer(n * 112 * 112 * 192 * 2, zero=False)
    timed("conv2/norm2 alone", lambda: L.check(lib.fcn_lrn_fwd_f16(x.ptr, y.ptr, n * 112 * 112, 192, 192, 192, 5, 1e-4, 0.75, 1.0, None)), 2 * n * 112 * 112 * 192 * 2)
    x.free(); y.free()
    # the nine 3x3 / stride 1 inception poolings and pool3 (3x3 / stride 2, 56 -> 28)
    for name, h, c in (("inception_3a/pool", 56, 192), ("inception_3b/pool", 56, 256), ("inception_4a/pool", 28, 480), ("inception_4b/pool", 28, 512),
                       ("inception_4c/pool", 28, 512), ("inception_4d/pool", 28, 512), ("inception_4e/pool", 28, 528), ("inception_5a/pool", 28, 832),
                       ("inception_5b/pool", 28, 832)):
        x, y = blob(h, h, c), DeviceBuffer(n * h * h * c * 2, zero=False)
        total += timed(name, lambda: L.check(lib.fcn_maxpool_fwd_f16(x.ptr, y.ptr, n, h, h, c, c, 3, 1, 1, h, h, c, 0, None)), 2 * n * h * h * c * 2)
        x.free(); y.free()
    x, y = blob(56, 56, 480), DeviceBuffer(n * 28 * 28 * 480 * 2, zero=False)
    total += timed("pool3/3x3_s2", lambda: L.check(lib.fcn_maxpool_fwd_f16(x.ptr, y.ptr, n, 56, 56, 480, 480, 3, 2, 0, 28, 28, 480, 0, None)),
                   n * (56 * 56 + 28 * 28) * 480 * 2)
    print("sum of the net's launches: %.1f us" % total)


if __name__ == "__main__":
    main()
