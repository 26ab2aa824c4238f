#!/usr/bin/env python
"""GPU diagnostic: where the waves of conv_stream_f16 spend a launch, from 100 MHz clock sums taken inside the kernel.

Needs the stamped build:  make -C fcn_object_detector_amd/csrc exp EXP=-DFCN_STREAM_STAMPS EXPNAME=sstamps
    FCN_LIB_PATH=fcn_object_detector_amd/libfcnhip_sstamps.so SWEEP_F16=1 SWEEP_BATCH=32 SWEEP_CFGS=36 python tools/stream_timeline.py conv2_3x3

Per shape and configuration, medians over workgroups (microseconds): multiplying wave 0 - chunk bodies, barrier waits, epilogues,
bookkeeping, next-tile set-up, whole kernel; slab-loading wave / weight-loading wave - issue work, waits for loads, barrier waits."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fcn_object_detector_amd import lib as L  # noqa: E402
from fcn_object_detector_amd.engine import DeviceBuffer  # noqa: E402
from gpu_util import conv_desc, dev_from  # noqa: E402
from conv_sweep import SHAPES  # noqa: E402

BATCH = int(os.environ.get("SWEEP_BATCH", "32"))
CFGS = [int(c) for c in os.environ.get("SWEEP_CFGS", "36").split(",") if c]


def main():
    want = sys.argv[1:] or ["conv2_3x3"]
    L.call("fcn_init", 0)
    lib = L.load()
    lib.fcn_debug_stream_stamps.restype = C.c_int
    lib.fcn_debug_stream_stamps.argtypes = [C.c_void_p, C.c_int]
    stamps = DeviceBuffer(256 * 3 * 8 * 8)
    rng = np.random.default_rng(0)
    for name, probs in SHAPES:
        if name not in want:
            continue
        keep, descs = [], []
        for (cin, cout, k, pad, s, h, w) in probs:
            cin = (cin + 7) // 8 * 8
            co = (cout + 7) // 8 * 8
            x = dev_from(rng.standard_normal((BATCH, h, w, cin)).astype(np.float16))
            wt = dev_from((rng.standard_normal((cout, k, k, cin)) * 0.05).astype(np.float16))
            b = dev_from(np.zeros(cout, np.float32))
            y = dev_from(np.zeros((BATCH, h, w, co), np.float16))
            keep += [x, wt, b, y]
            descs.append(conv_desc(x, wt, b, y, BATCH, h, w, cin, cin, cout, k, pad, s, h, w, co, 0, L.CONV_RELU | L.CONV_F16))
        arr = (L.ConvDesc * len(descs))(*descs)
        ws = DeviceBuffer(int(lib.fcn_conv2d_group_workspace_bytes(len(descs))), zero=False)
        for cfg in CFGS:
            grp = L.ConvGroup()
            if lib.fcn_conv2d_group_prepare(arr, len(descs), ws.ptr, cfg, C.byref(grp)) != 0:
                continue
            L.check(lib.fcn_debug_stream_stamps(None, 0))
            for _ in range(3):
                L.call("fcn_conv2d_fwd_group_f32", C.byref(grp), None)
            L.call("fcn_memset_async", stamps.ptr, 0, stamps.nbytes, None)
            L.call("fcn_device_sync")
            L.check(lib.fcn_debug_stream_stamps(stamps.ptr, 256))
            L.call("fcn_conv2d_fwd_group_f32", C.byref(grp), None)
            L.call("fcn_device_sync")
            L.check(lib.fcn_debug_stream_stamps(None, 0))
            raw = np.empty((256, 3, 8), np.uint64)
            L.call("fcn_memcpy_d2h_async", raw.ctypes.data, stamps.ptr, raw.nbytes, None)
            L.call("fcn_device_sync")
            us = raw.astype(np.float64) / 100.0
            ok = us[:, 0, 7] > 0
            m = np.median(us[ok], axis=0)
            m[0, 6] = 0.0
            print("== %s cfg %d  %d tiles, %d workgroups stamped" % (name, cfg, grp.total_tiles, int(ok.sum())))
            clk = raw[ok, 0, 6].astype(np.float64) / np.maximum(raw[ok, 0, 7].astype(np.float64) * 10.0, 1.0)      # shader cycles / ns
            print("   multiplier: chunks %.1f  barrier %.1f  epilogue %.1f  bookkeeping %.1f  next-tile %.1f  | kernel %.1f us (max %.1f)  shader clock %.2f GHz" % (
                m[0, 0], m[0, 1], m[0, 2], m[0, 3], m[0, 4], m[0, 7], us[ok, 0, 7].max(), float(np.median(clk))))
            print("   slab wave : issue %.1f  load wait %.1f  barrier %.1f" % (m[1, 0], m[1, 1], m[1, 2]))
            print("   weight wave: issue %.1f  load wait %.1f  barrier %.1f" % (m[2, 0], m[2, 1], m[2, 2]), flush=True)


if __name__ == "__main__":
    main()
