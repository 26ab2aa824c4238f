#!/usr/bin/env python
"""GPU: where a `caffe train` iteration with the device-rendered data layer spends its time (host planning vs device step)."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fcn_object_detector_amd", "python"))
from fcn_object_detector_amd import lib as L, models  # noqa: E402
from fcn_object_detector_amd.solver import Solver  # noqa: E402

os.environ["FCN_DATA_SEED"] = "1"
with tempfile.TemporaryDirectory() as tmp:
    net = os.path.join(tmp, "t.prototxt")
    open(net, "w").write(models.googlenet_detectnet_train("data_argumentation_layer", "DataArgumentationLayer",
                                                         "448,448,16,1,8,synthetic:1,detectnet", num_classes=1))
    sol = os.path.join(tmp, "s.prototxt")
    open(sol, "w").write('net: "%s"\nbase_lr: 1e-4\nmomentum: 0.9\nlr_policy: "fixed"\ndisplay: 0\nmax_iter: 100000\nsnapshot: 0\n' % net)
    s = Solver(sol, device=0, log=None)
    s.step(3)
    tf = ts = 0.0
    n = 20
    for _ in range(n):
        t0 = time.perf_counter()
        s._feed()
        L.call("fcn_stream_sync", s.engine.stream)
        t1 = time.perf_counter()
        s.engine.step()
        t2 = time.perf_counter()
        tf += t1 - t0
        ts += t2 - t1
    print("feed (plan + render, synced) %.3f ms   step %.3f ms" % (tf / n * 1e3, ts / n * 1e3))
    lay = s.py_layers[0][1]
    t0 = time.perf_counter()
    for _ in range(200):
        lay.plan_scene()
    print("plan_scene %.3f ms/sample" % ((time.perf_counter() - t0) / 200 * 1e3))
    plan = lay.plan_scene()
    t0 = time.perf_counter()
    for _ in range(200):
        lay._renderer.render(0, plan)
    t1 = time.perf_counter()
    L.call("fcn_stream_sync", s.engine.stream)
    t2 = time.perf_counter()
    print("render enqueue %.3f ms/sample, device drain of 200 renders %.3f ms (%.3f ms each)" % ((t1 - t0) / 200 * 1e3, (t2 - t0) * 1e3, (t2 - t0) / 200 * 1e3))
    for blur in (dict(kind="gauss", sigma=3.0), dict(kind="gauss", sigma=0.8), dict(kind="box", k=7), dict(kind="median", k=3),
                 dict(kind="median", k=5), dict(kind="median", k=7), None):
        p = dict(plan)
        p["view"] = None
        p["color"] = dict(plan["color"], blur=blur) if blur else None
        lay._renderer.render(0, p)
        L.call("fcn_stream_sync", s.engine.stream)
        t0 = time.perf_counter()
        for _ in range(100):
            lay._renderer.render(0, p)
        L.call("fcn_stream_sync", s.engine.stream)
        print("render with %-40s %.3f ms/sample (enqueue + drain)" % (blur, (time.perf_counter() - t0) / 100 * 1e3))
    s.step(20, pipeline=True) if "pipeline" in s.step.__code__.co_varnames else None
    t0 = time.perf_counter()
    s.step(50, pipeline=True)
    L.call("fcn_stream_sync", s.engine.stream)
    print("pipelined solver iteration %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
    # where the pipelined iteration spends its host time
    tb = tf = te = 0.0
    n = 30
    s._feed()
    for _ in range(n):
        t0 = time.perf_counter()
        s.engine.step_begin()
        t1 = time.perf_counter()
        s._feed()
        t2 = time.perf_counter()
        s.engine.step_end()
        t3 = time.perf_counter()
        tb += t1 - t0
        tf += t2 - t1
        te += t3 - t2
    print("pipelined loop: step_begin %.3f ms, feed %.3f ms, step_end (wait) %.3f ms, total %.3f ms" % (tb / n * 1e3, tf / n * 1e3, te / n * 1e3, (tb + tf + te) / n * 1e3))
    tb = te = 0.0
    for _ in range(n):
        t0 = time.perf_counter()
        s.engine.step_begin()
        t1 = time.perf_counter()
        s.engine.step_end()
        t3 = time.perf_counter()
        tb += t1 - t0
        te += t3 - t1
    print("no feed:        step_begin %.3f ms, step_end (wait) %.3f ms, total %.3f ms" % (tb / n * 1e3, te / n * 1e3, (tb + te) / n * 1e3))
