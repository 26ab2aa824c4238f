#!/usr/bin/env python
"""GPU experiment: LDS canary workgroups on one stream, library kernels on another.  Which co-runner (if any) modifies LDS that
belongs to a different workgroup?"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fcn_object_detector_amd import lib as L  # noqa: E402
from fcn_object_detector_amd.engine import DeviceBuffer  # noqa: E402
from gpu_util import conv_desc, dev_from  # noqa: E402

lib = L.load()
L.call("fcn_init", 0)
can = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcanary.so"))
can.canary_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
sa, sb = C.c_void_p(), C.c_void_p()
L.call("fcn_stream_create", C.byref(sa))
L.call("fcn_stream_create", C.byref(sb))
out = DeviceBuffer(4 * 256, zero=True)
rng = np.random.default_rng(0)

# co-runner problems: a convolution with K-split tiles, and its weight gradient
n, h, w, cin, cout, k = 8, 28, 28, 160, 320, 3
x = dev_from(rng.standard_normal((n, h, w, cin)).astype(np.float32))
wt = dev_from((rng.standard_normal((cout, k, k, cin)) * 0.05).astype(np.float32))
y = dev_from(np.zeros((n, h, w, cout), np.float32))
d = conv_desc(x, wt, None, y, n, h, w, cin, cin, cout, k, 1, 1, h, w, cout)
dw = DeviceBuffer(cout * k * k * cin * 4 + cout * 4)
ws_by_cfg = {}
for c in (0, 1, 2, 3):          # the pixel split (and with it the workspace) depends on the tile shape: size it per forced shape
    os.environ["FCN_WGRAD_CFG"] = str(c)
    ws_by_cfg[c] = DeviceBuffer(int(lib.fcn_conv2d_wgrad_workspace_floats(C.byref(d), None)) * 4 + 4096)


def conv_corunner(cfg):
    os.environ["FCN_CONV_CFG"] = str(cfg)
    L.check(lib.fcn_conv2d_fwd_f32(C.byref(d), sb))


def wgrad_corunner(cfg):
    os.environ["FCN_WGRAD_CFG"] = str(cfg)
    L.check(lib.fcn_conv2d_wgrad_f32(C.byref(d), dw.ptr, None, ws_by_cfg[cfg].ptr, sb))


def trial(name, fn, kb, reps=300):
    L.call("fcn_memset_async", out.ptr, 0, out.nbytes, None)
    L.call("fcn_device_sync")
    for _ in range(reps):
        can.canary_launch(512, kb, 40, out.ptr, sa)
        if fn:
            for _ in range(3):
                fn()
    L.call("fcn_device_sync")
    host = np.zeros(256, np.uint32)
    L.call("fcn_memcpy_d2h_async", host.ctypes.data, out.ptr, host.nbytes, None)
    L.call("fcn_device_sync")
    log = [(int(host[1 + 2 * i]), hex(int(host[2 + 2 * i]))) for i in range(min(int(host[0]), 6))]
    print("%-28s canary %3d KiB: %d foreign words %s" % (name, kb, int(host[0]), log))


for kb in (16, 48, 64):
    trial("alone", None, kb)
    for cfg in (5, 14, 21):
        trial("beside conv cfg%d" % cfg, lambda c=cfg: conv_corunner(c), kb)
    for cfg in (0, 1, 2, 3):
        trial("beside wgrad cfg%d" % cfg, lambda c=cfg: wgrad_corunner(c), kb)
