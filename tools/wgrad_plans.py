#!/usr/bin/env python
"""CPU: print the weight-gradient planner's decisions (FCN_WGRAD_PLAN_LOG) for the sweep's layer shapes; no GPU needed."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("FCN_QUIET", "1")
os.environ["FCN_WGRAD_PLAN_LOG"] = "1"
os.environ.setdefault("FCN_WGRAD_CFG", "4")
from fcn_object_detector_amd import lib as L  # noqa: E402
from gpu_util import conv_desc  # noqa: E402
from wgrad_sweep import SHAPES  # noqa: E402


class _Fake:
    ptr = 0x1000


def main():
    lib = L.load()
    for name, cin, cout, k, pad, s, h, w, n in SHAPES:
        oh, ow = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
        d = conv_desc(_Fake, _Fake, None, _Fake, n, h, w, cin, cin, cout, k, pad, s, oh, ow, (cout + 3) // 4 * 4, 0)
        sp = C.c_int(0)
        sys.stderr.write("%-13s " % name)
        sys.stderr.flush()
        lib.fcn_conv2d_wgrad_workspace_floats(C.byref(d), C.byref(sp))


if __name__ == "__main__":
    main()
