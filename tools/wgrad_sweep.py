#!/usr/bin/env python
"""GPU micro-benchmark: the weight-gradient kernel across its tile shapes (FCN_WGRAD_CFG) on training-net layer shapes."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fcn_object_detector_amd import lib as L  # noqa: E402
from fcn_object_detector_amd.engine import DeviceBuffer  # noqa: E402
from gpu_util import conv_desc, dev_from, dev_to  # noqa: E402

SHAPES = [  # name, cin, cout, k, pad, stride, h, w, n
    ("conv1", 4, 64, 7, 3, 2, 448, 448, 8),
    ("conv2_3x3", 64, 192, 3, 1, 1, 112, 112, 8),
    ("3b_3x3", 128, 192, 3, 1, 1, 56, 56, 8),
    ("3a_pool_proj", 192, 32, 1, 0, 1, 56, 56, 8),
    ("4a_1x1", 480, 192, 1, 0, 1, 28, 28, 8),
    ("4d_3x3", 144, 288, 3, 1, 1, 28, 28, 8),
    ("5b_3x3", 192, 384, 3, 1, 1, 28, 28, 8),
    ("5b_1x1", 832, 384, 1, 0, 1, 28, 28, 8),
    ("3a_5x5", 16, 32, 5, 2, 1, 56, 56, 8),
    ("3b_5x5", 32, 96, 5, 2, 1, 56, 56, 8),
    ("4e_3x3", 160, 320, 3, 1, 1, 28, 28, 8),
    ("4b_5x5r", 512, 24, 1, 0, 1, 28, 28, 8),
    ("3a_3x3", 96, 128, 3, 1, 1, 56, 56, 8),
    ("4a_3x3", 96, 208, 3, 1, 1, 28, 28, 8),
]


def main():
    L.call("fcn_init", 0)
    lib = L.load()
    sp = C.c_void_p()
    L.call("fcn_stream_create", C.byref(sp))
    st = sp.value
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.call("fcn_event_create", C.byref(e0))
    L.call("fcn_event_create", C.byref(e1))
    rng = np.random.default_rng(0)
    for name, cin, cout, k, pad, s, h, w, n in SHAPES:
        if sys.argv[1:] and name not in sys.argv[1:]:
            continue
        oh, ow = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
        x = dev_from(rng.standard_normal((n, h, w, cin)).astype(np.float32))
        co4 = (cout + 3) // 4 * 4
        dy = dev_from(rng.standard_normal((n, oh, ow, co4)).astype(np.float32))
        d = conv_desc(x, x, None, dy, n, h, w, cin, cin, cout, k, pad, s, oh, ow, co4, 0)
        flops = 2.0 * n * oh * ow * cout * cin * k * k
        dw = dev_from(np.zeros((cout, k, k, cin), np.float32))
        db = dev_from(np.zeros(cout, np.float32))
        dbp = None if os.environ.get("SWEEP_NO_BIAS") else db.ptr
        line = "%-14s %6.3f GFLOP |" % (name, flops / 1e9)
        ref = None
        for cfg in [None] + os.environ.get("SWEEP_CFGS", "0,1,2,3,4").split(","):
            if cfg is None:
                os.environ.pop("FCN_WGRAD_CFG", None)
            else:
                os.environ["FCN_WGRAD_CFG"] = cfg
            splits = C.c_int(0)
            nfl = int(lib.fcn_conv2d_wgrad_workspace_floats(C.byref(d), C.byref(splits)))
            ws = DeviceBuffer(nfl * 4, zero=False)
            for _ in range(2):
                L.call("fcn_conv2d_wgrad_f32", C.byref(d), dw.ptr, dbp, ws.ptr, st)
            L.call("fcn_event_record", e0, st)
            reps = 10
            for _ in range(reps):
                L.call("fcn_conv2d_wgrad_f32", C.byref(d), dw.ptr, dbp, ws.ptr, st)
            L.call("fcn_event_record", e1, st)
            L.call("fcn_event_sync", e1)
            ms = C.c_float()
            L.call("fcn_event_elapsed_ms", e0, e1, C.byref(ms))
            us = ms.value / reps * 1e3
            got = dev_to(dw, (cout, k, k, cin))
            if ref is None:
                ref = got
            err = float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))
            line += " %s/s%d %7.1fus %5.1fTF e%.0e |" % (cfg or "A", splits.value, us, flops / us / 1e6, err)
        print(line, flush=True)


if __name__ == "__main__":
    main()
