#!/usr/bin/env python
"""GPU: does any weight-gradient launch write past the workspace the planner sized?  The workspace is re-allocated with a guard
zone behind it, the guard is filled with a pattern, a few training steps run, and the guard is read back."""
import os
import random
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fcn_object_detector_amd", "python"))
from fcn_object_detector_amd import lib as L, models  # noqa: E402
from fcn_object_detector_amd.engine import DeviceBuffer  # noqa: E402
from fcn_object_detector_amd.solver import Solver  # noqa: E402

tmp = tempfile.mkdtemp()
net = os.path.join(tmp, "t.prototxt")
shape = sys.argv[1] if len(sys.argv) > 1 else "128,96,16,2,2"
open(net, "w").write(models.googlenet_detectnet_train("data_argumentation_layer", "DataArgumentationLayer", shape + ",synthetic:2,detectnet",
                                                     num_classes=int(shape.split(",")[3])))
sol = os.path.join(tmp, "s.prototxt")
open(sol, "w").write('net: "%s"\nbase_lr: 1e-4\nmomentum: 0.9\nlr_policy: "fixed"\ndisplay: 0\nmax_iter: 100\nsnapshot: 0\n' % net)
os.environ["FCN_WGRAD_STREAM"] = "0"
s = Solver(sol, device=0, log=None, autotune=False)
eng = s.engine
old = eng._ws
guard = 4 << 20
print("workspace: %d bytes" % old.nbytes)
# the ops captured the old buffer's address through self._ws.ptr at call time (attribute lookup), so swapping the object is enough
new = DeviceBuffer(old.nbytes + guard, zero=False)
L.call("fcn_memset_async", new.ptr, 0x5A, new.nbytes, None)
L.call("fcn_device_sync")
eng._ws = new
random.seed(1)
s.step(2)
back = np.empty(guard, np.uint8)
L.call("fcn_memcpy_d2h_async", back.ctypes.data, new.ptr + old.nbytes, guard, None)
L.call("fcn_device_sync")
bad = np.nonzero(back != 0x5A)[0]
print("guard bytes modified: %d%s" % (bad.size, (" first at +%d, last at +%d" % (bad[0], bad[-1])) if bad.size else ""))
