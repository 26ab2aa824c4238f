#!/usr/bin/env python
"""GPU diagnostic: when the workgroups of conv_wgrad_split_kernel ran and what their waves waited for.

Needs the stamped build:  make -C fcn_object_detector_amd/csrc exp EXP=-DFCN_WS_STAMPS EXPNAME=ws_stamps EXPSRC=train
    FCN_LIB_PATH=fcn_object_detector_amd/libfcnhip_ws_stamps.so FCN_WGRAD_CFG=4 python tools/wgrad_timeline.py conv2_3x3

Per shape: launch span, workgroups per round of the chip, per class of workgroup (accumulators of wave 0) the shader cycles per
chunk, the share of them multiplying wave 0 spent at barriers and staging wave 0 spent waiting for loads, and the clock held."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fcn_object_detector_amd import lib as L  # noqa: E402
from fcn_object_detector_amd.engine import DeviceBuffer  # noqa: E402
from gpu_util import conv_desc, dev_from, dev_to  # noqa: E402
from wgrad_sweep import SHAPES  # noqa: E402


def main():
    L.call("fcn_init", 0)
    lib = L.load()
    cap = 8192
    lib.fcn_debug_wgrad_stamps.argtypes = [C.c_void_p, C.c_int]
    rng = np.random.default_rng(0)
    for name, cin, cout, k, pad, s, h, w, n in SHAPES:
        if sys.argv[1:] and name not in sys.argv[1:]:
            continue
        oh, ow = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
        stamps = DeviceBuffer(cap * 8 * 8, zero=True)      # a fresh (zeroed) stamp buffer per shape
        assert lib.fcn_debug_wgrad_stamps(stamps.ptr, cap) == 0
        x = dev_from(rng.standard_normal((n, h, w, cin)).astype(np.float32))
        co4 = (cout + 3) // 4 * 4
        dy = dev_from(rng.standard_normal((n, oh, ow, co4)).astype(np.float32))
        d = conv_desc(x, x, None, dy, n, h, w, cin, cin, cout, k, pad, s, oh, ow, co4, 0)
        dw = dev_from(np.zeros((cout, k, k, cin), np.float32))
        db = dev_from(np.zeros(cout, np.float32))
        splits = C.c_int(0)
        nfl = int(lib.fcn_conv2d_wgrad_workspace_floats(C.byref(d), C.byref(splits)))
        ws = DeviceBuffer(nfl * 4, zero=False)
        e0, e1 = C.c_void_p(), C.c_void_p()
        L.call("fcn_event_create", C.byref(e0))
        L.call("fcn_event_create", C.byref(e1))
        for _ in range(3):
            L.call("fcn_event_record", e0, None)
            L.call("fcn_conv2d_wgrad_f32", C.byref(d), dw.ptr, db.ptr, ws.ptr, None)
            L.call("fcn_event_record", e1, None)
        L.call("fcn_device_sync")
        ms = C.c_float()
        L.call("fcn_event_elapsed_ms", e0, e1, C.byref(ms))
        print("%s: events around the call (kernel + reduction): %.1f us" % (name, ms.value * 1e3))
        st = dev_to(stamps, (cap, 8), np.uint64)
        st = st[st[:, 1] > 0]
        t0 = st[:, 0].min()
        beg, end = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0      # microseconds
        flops = 2.0 * n * oh * ow * cout * cin * k * k
        span = end.max()
        print("%s: %d workgroups, splits %d, span %.1f us (%.1f TF/s without the reduction)" % (name, len(st), splits.value, span, flops / span / 1e6))
        order = np.argsort(beg)
        live = [(b, 1) for b in beg] + [(e, -1) for e in end]
        live.sort()
        cur, tl = 0, []
        for t, dlt in live:
            cur += dlt
            tl.append((t, cur))
        for frac in (0.1, 0.25, 0.5, 0.75, 0.9, 0.97):
            t = frac * span
            print("   at %5.1f us: %3d workgroups resident" % (t, max([c for tt, c in tl if tt <= t][-1:] or [0])))
        cnt = (st[:, 5] & 0xff).astype(int)
        for c in sorted(set(cnt)):
            m = cnt == c
            nch = st[m, 4].astype(float)
            cyc = st[m, 2].astype(float)
            dur = (end - beg)[m]
            print("   wave-0 accumulators %d: %4d wgs, chunks %3d, %7.0f cycles/chunk (ideal %d at 32-pixel chunks), barrier wait %4.1f%%, staging load wait %4.1f%%, "
                  "store %5.0f cycles, clock %.2f GHz, wg %.1f us" %
                  (c, m.sum(), np.median(nch), np.median(cyc / np.maximum(nch, 1)), 1024 * c, 100 * np.median(st[m, 6] / np.maximum(cyc, 1)),
                   100 * np.median(st[m, 7] / np.maximum(cyc, 1)), np.median(st[m, 3]), np.median((cyc + st[m, 3]) / dur / 1e3), np.median(dur)))
        xcc = ((st[:, 5] >> 8) & 15).astype(int)
        print("   workgroups per XCC id:", dict(zip(*np.unique(xcc, return_counts=True))))


if __name__ == "__main__":
    main()
