#!/usr/bin/env python
"""GPU: which gradient BLOB (data-gradient chain) differs first when the two-stream step is not deterministic."""
import os
import random
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fcn_object_detector_amd", "python"))
from fcn_object_detector_amd import lib as L, models  # noqa: E402
from fcn_object_detector_amd.solver import Solver  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
tmp = tempfile.mkdtemp()
net = os.path.join(tmp, "t.prototxt")
open(net, "w").write(models.googlenet_detectnet_train("data_argumentation_layer", "DataArgumentationLayer", "128,96,16,2,2,synthetic:2,detectnet",
                                                     num_classes=2))
sol = os.path.join(tmp, "s.prototxt")
open(sol, "w").write('net: "%s"\nbase_lr: 1e-4\nmomentum: 0.9\nweight_decay: 1e-6\nlr_policy: "fixed"\ndisplay: 0\nmax_iter: 100\nsnapshot: 0\n' % net)


def run():
    s = Solver(sol, device=0, log=None, autotune=False)
    lay = s.py_layers[0][1]
    random.seed(5)
    lay._color_rng = np.random.default_rng(1234)
    s.step(1)
    eng = s.engine
    out = {}
    seen = set()
    for name, g in eng.grad_blobs.items():
        if g.buf.ptr in seen or len(g.shape) != 4:
            continue
        seen.add(g.buf.ptr)
        a = np.empty(g.buf.nbytes // 4, np.float32)
        L.call("fcn_memcpy_d2h_async", a.ctypes.data, g.buf.ptr, a.nbytes, eng.stream)
        out[name] = (a, g.cstride)
    L.call("fcn_stream_sync", eng.stream)
    order = [l.tops[0] for l in eng.spec.layers if l.tops and l.tops[0] in out]
    s.close()
    return out, order


os.environ["FCN_WGRAD_STREAM"] = "0"
ref, order = run()
os.environ["FCN_WGRAD_STREAM"] = "1"
for i in range(reps):
    got, _ = run()
    bad = [k for k in order if not np.array_equal(got[k][0], ref[k][0])]
    if bad:
        k = bad[-1]                      # deepest blob in net order = first touched by backward
        a, cs = got[k]
        d = (a != ref[k][0]).reshape(-1, cs)
        ch = np.nonzero(d.any(0))[0]
        px = np.nonzero(d.any(1))[0]
        print("run %d: %d blobs differ; deepest: %s, %d channels %s..., %d of %d pixels (first %s)" % (
            i, len(bad), k, len(ch), ch[:24].tolist(), len(px), d.shape[0], px[:6].tolist()))
