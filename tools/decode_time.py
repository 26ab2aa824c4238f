#!/usr/bin/env python
"""GPU: duration of the decode + groupRectangles launch on the maps of the random-weight deploy net (worst case: one class fires on
every cell) and on sparse synthetic maps."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fcn_object_detector_amd import lib as L, models, proto  # noqa: E402
from fcn_object_detector_amd.detector import FCNObjectDetector, HeadMapping  # noqa: E402
from fcn_object_detector_amd.engine import Engine  # noqa: E402
from fcn_object_detector_amd.netspec import NetSpec, fill_params  # noqa: E402

msg = proto.parse_text(models.googlenet_detectnet_deploy(1, 448, 448, 4))
spec = NetSpec(msg, "TEST")
spec.infer()
eng = Engine(NetSpec(msg, "TEST"), params=fill_params(spec, seed=1234), device=0, autotune=False)
d = FCNObjectDetector(eng, mapping=HeadMapping.detectnet_deploy())
frame = np.random.default_rng(0).integers(0, 256, (480, 640, 3), dtype=np.uint8)
d.run_detector(frame)
e0, e1 = C.c_void_p(), C.c_void_p()
L.call("fcn_event_create", C.byref(e0))
L.call("fcn_event_create", C.byref(e1))
cvg = eng.read_blob("coverage")
print("cells above 0.5 per class:", (cvg[0] >= 0.5).reshape(4, -1).sum(1))
L.call("fcn_event_record", e0, eng.stream)
for _ in range(50):
    d.decoder.launch(*d._cvg_args, *d._box_args, eng.stream)
L.call("fcn_event_record", e1, eng.stream)
L.call("fcn_event_sync", e1)
ms = C.c_float()
L.call("fcn_event_elapsed_ms", e0, e1, C.byref(ms))
print("decode + groupRectangles launch: %.1f us" % (ms.value / 50 * 1e3))
