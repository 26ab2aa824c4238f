#!/usr/bin/env python
"""GPU micro-benchmark: cost of one dependent tiny kernel launch in a stream and inside a hipGraph (run on the GPU box)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fcn_object_detector_amd import lib as L  # noqa: E402
from gpu_util import dev_from  # noqa: E402


def main():
    L.call("fcn_init", 0)
    sp = C.c_void_p()
    L.call("fcn_stream_create", C.byref(sp))
    st = sp.value
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.call("fcn_event_create", C.byref(e0))
    L.call("fcn_event_create", C.byref(e1))
    x = dev_from(np.ones(256, np.float32))
    y = dev_from(np.zeros(256, np.float32))
    ms = C.c_float()
    for n in (64, 1 << 16):
        for _ in range(10):
            L.call("fcn_relu_fwd_f32", x.ptr, y.ptr, 64, 0.0, st)
        reps = 200
        L.call("fcn_event_record", e0, st)
        for _ in range(reps):
            L.call("fcn_relu_fwd_f32", x.ptr, y.ptr, 64, 0.0, st)
        L.call("fcn_event_record", e1, st)
        L.call("fcn_event_sync", e1)
        L.call("fcn_event_elapsed_ms", e0, e1, C.byref(ms))
        print("stream: tiny relu launch %.2f us each" % (ms.value / reps * 1e3))
    # inside a graph
    L.call("fcn_graph_begin", st)
    for _ in range(200):
        L.call("fcn_relu_fwd_f32", x.ptr, y.ptr, 64, 0.0, st)
    g = C.c_void_p()
    L.call("fcn_graph_end", st, C.byref(g))
    for _ in range(3):
        L.call("fcn_graph_launch", g, st)
    L.call("fcn_event_record", e0, st)
    for _ in range(5):
        L.call("fcn_graph_launch", g, st)
    L.call("fcn_event_record", e1, st)
    L.call("fcn_event_sync", e1)
    L.call("fcn_event_elapsed_ms", e0, e1, C.byref(ms))
    print("graph: tiny relu node %.2f us each" % (ms.value / 1000 * 1e3))


if __name__ == "__main__":
    main()
