"""End-to-end parity of the DetectNet GoogLeNet forward (reference models/deploy.prototxt) on the GPU vs the CPU oracle,
through the engine and through the pycaffe-compatible front end (run with -m gpu)."""
import os
import sys

import numpy as np
import pytest

from conftest import elem_err, PYCAFFE, rel_err
from fcn_object_detector_amd import models, proto
from fcn_object_detector_amd.engine import Engine
from fcn_object_detector_amd.netspec import NetSpec, fill_params
from oracle import detect_ref as D
from oracle.net_ref import RefNet

pytestmark = pytest.mark.gpu
TOL = 1e-3          # north_star: 1e-3 relative fp32


def build(batch, h, w, classes, seed=1234, **kw):
    msg = proto.parse_text(models.googlenet_detectnet_deploy(batch, h, w, classes))
    spec = NetSpec(msg, "TEST")
    spec.infer()
    params = fill_params(spec, seed=seed)
    return msg, params, Engine(NetSpec(msg, "TEST"), params=params, device=0, **kw)


def oracle_forward(msg, params, x):
    ref = RefNet(msg, "TEST", params)
    ref.blobs["data"] = x
    return ref.forward()


def test_deploy_448_matches_oracle(gpu):
    """BASELINE config 1/2: seeded fillers (default_rng(1234)), random uint8 frame (default_rng(0)) through the
    reference pre-processing; coverage/bboxes and a sample of intermediate blobs within 1e-3 relative."""
    msg, params, eng = build(1, 448, 448, 4)
    frame = np.random.default_rng(0).integers(0, 256, (448, 448, 3), dtype=np.uint8)
    x = D.preprocess_frame(frame, 448, 448)[None]
    eng.host_array("data")[...] = x
    out = eng.forward()
    rb = oracle_forward(msg, params, x)
    assert out["coverage"].shape == (1, 4, 28, 28) and out["bboxes"].shape == (1, 16, 28, 28)
    for name in ("coverage", "bboxes"):
        assert rel_err(out[name], rb[name]) < TOL, name
        worst, at = elem_err(out[name], rb[name], TOL)      # and element by element: |a - b| <= 1e-3 |b| + 1e-3 rms(b)
        assert worst <= 1.0, (name, worst, at)
    for name in ("transformed_data", "conv1/7x7_s2", "pool1/norm1", "conv2/norm2", "pool2/3x3_s2", "inception_3a/pool",
                 "inception_3a/output", "inception_3b/5x5", "pool3/3x3_s2", "inception_4a/3x3_reduce", "inception_4e/output",
                 "inception_5b/output", "pool5/drop_s1", "cvg/classifier"):
        assert rel_err(eng.read_blob(name), rb[name]) < TOL, name
    # eager launches and graph replay give identical bits; replay is deterministic
    again = {k: v.copy() for k, v in eng.forward().items()}
    eager = eng.forward(use_graph=False)
    for name in again:
        assert np.array_equal(again[name], out[name]) and np.array_equal(eager[name], out[name])
    eng.close()


@pytest.mark.parametrize("batch,hw", [(1, (448, 448)), (2, (96, 128))])
def test_heads_as_the_tail_of_inception_5b(gpu, monkeypatch, batch, hw):
    """FCN_CONV_TAIL=1 (opt-in: measured no faster at batch 1): cvg/classifier + bbox/regressor evaluated by the launches that
    write inception_5b/output instead of a launch of their own - the plan holds no heads launch, the heads' blobs match the oracle and
    the default plan, graph replay and eager launches give the same bits."""
    h, w = hw
    cls = 4 if batch == 1 else 1      # (one class: cvg/classifier has a single channel - not in whole fours - and the plan keeps the heads' launch)
    x = np.random.default_rng(11).random((batch, 3, h, w), dtype=np.float32) * 255 - 127
    msg, params, plain = build(batch, h, w, cls)
    plain.host_array("data")[...] = x
    want = {k: v.copy() for k, v in plain.forward().items()}
    plain.close()
    monkeypatch.setenv("FCN_CONV_TAIL", "1")
    msg, params, eng = build(batch, h, w, cls)
    names = [op.name for op in eng.ops]
    if cls == 4:
        assert not any(nm.startswith("cvg/classifier") for nm in names) and any("tail: cvg/classifier+bbox/regressor" in nm for nm in names)
    else:
        assert any(nm.startswith("cvg/classifier") for nm in names)
    eng.host_array("data")[...] = x
    out = {k: v.copy() for k, v in eng.forward().items()}
    rb = oracle_forward(msg, params, x)
    for name in ("coverage", "bboxes"):
        assert rel_err(out[name], rb[name]) < TOL and rel_err(out[name], want[name]) < 1e-5, name
        assert elem_err(out[name], rb[name], TOL)[0] <= 1.0
    assert rel_err(eng.read_blob("cvg/classifier"), rb["cvg/classifier"]) < TOL and rel_err(eng.read_blob("inception_5b/output"), rb["inception_5b/output"]) < TOL
    again = eng.forward()
    eager = eng.forward(use_graph=False)
    for name in out:
        assert np.array_equal(again[name], out[name]) and np.array_equal(eager[name], out[name])
    eng.close()


@pytest.mark.parametrize("fuse,group", [(False, False), (True, False)])
def test_unfused_plans_agree(gpu, fuse, group):
    msg, params, eng = build(2, 96, 128, 3, fuse=fuse, group_convs=group)
    x = np.random.default_rng(3).random((2, 3, 96, 128), dtype=np.float32)
    eng.host_array("data")[...] = x
    out = eng.forward()
    rb = oracle_forward(msg, params, x)
    for name in ("coverage", "bboxes"):
        assert rel_err(out[name], rb[name]) < TOL
    eng.close()


def test_batch_and_odd_sizes(gpu):
    msg, params, eng = build(3, 80, 112, 1)
    x = np.random.default_rng(4).random((3, 3, 80, 112), dtype=np.float32)
    eng.host_array("data")[...] = x
    out = eng.forward()
    rb = oracle_forward(msg, params, x)
    assert out["coverage"].shape == (3, 1, 5, 7)
    for name in ("coverage", "bboxes"):
        assert rel_err(out[name], rb[name]) < TOL
    t = eng.forward_resident(3)
    assert t > 0
    eng.close()


def test_first_launch_of_every_kernel_happens_outside_the_stream_capture(gpu, monkeypatch, tmp_path):
    """An engine whose plan comes from the tune cache (no autotuning launches) must still run its launches ONCE eagerly before it
    captures them: code objects load lazily on a kernel's first launch, and a first launch inside a stream capture is what
    aborted under rocprofv3 in round 1.  Recorded: the order of graph begin and the first pass of every launch."""
    import json
    from fcn_object_detector_amd import lib as L
    msg = proto.parse_text(models.googlenet_detectnet_deploy(batch=1, height=96, width=128, num_classes=2))
    spec = NetSpec(msg, "TEST")
    spec.infer()
    params = fill_params(spec, seed=3)
    cache = tmp_path / "tune.json"
    monkeypatch.setenv("FCN_TUNE_CACHE", str(cache))
    eng = Engine(NetSpec(msg, "TEST"), params=params, device=0)      # fills the cache (autotuned)
    eng.close()
    assert len(json.load(open(cache))) > 5
    eng = Engine(NetSpec(msg, "TEST"), params=params, device=0)      # replays the cached plan: not a single launch so far
    events = []
    real_call = L.call

    def spy(name, *a):
        if name in ("fcn_graph_begin", "fcn_graph_end", "fcn_graph_launch"):
            events.append(name)
        return real_call(name, *a)

    for i, op in enumerate(eng.ops):
        op.run = (lambda st, f=op.run, i=i: (events.append(("op", i)), f(st))[1])
    # the graph behind forward() also holds the layout kernels of the upload and of the download (round 4: the download's kernel
    # used to be launched for the first time INSIDE the capture - nothing else runs it before the first forward()); the copies
    # themselves are plain async calls in front of and behind the graph launch, never graph nodes
    for nm in ("_upload_convert", "_upload_copy", "_download_copy"):
        real = getattr(eng, nm)
        setattr(eng, nm, (lambda name, st, real=real, nm=nm: (events.append((nm, name)), real(name, st))[1]))
    real_all = eng._download_convert_all      # (the layout kernels of all outputs: one launch for the two head blobs)
    eng._download_convert_all = lambda st: (events.append("_download_convert_all"), real_all(st))[1]
    monkeypatch.setattr(L, "call", spy)
    x = np.random.default_rng(0).random((1, 3, 96, 128), dtype=np.float32)
    eng.host_array("data")[...] = x
    out = {k: v.copy() for k, v in eng.forward().items()}
    begin, end = events.index("fcn_graph_begin"), len(events) - 1 - events[::-1].index("fcn_graph_end")
    n_ops = len(eng.ops)
    ops = [("op", i) for i in range(n_ops)]
    up = [("_upload_copy", "data"), ("_upload_convert", "data")]
    down = ["_download_convert_all"] + [("_download_copy", nm) for nm in eng.outputs]
    assert [e for e in events[:begin] if e != "fcn_graph_end"] == ops + up + down      # one eager pass of everything first
    assert events[begin + 1:end] == [("_upload_convert", "data")] + ops + ["_download_convert_all"]      # the capture: kernels only
    tail = [e for e in events[end + 1:] if e != "fcn_graph_launch"]
    assert tail == [("_upload_copy", "data")] + [("_download_copy", nm) for nm in eng.outputs]      # the frame itself: copy, graph, copies
    assert np.isfinite(out["coverage"]).all()
    again = eng.forward()                                   # a replay of the captured graph gives the same frame the same result
    assert all(np.array_equal(again[k], out[k]) for k in out)
    plain = eng.forward(use_graph=False)
    assert all(np.array_equal(plain[k], out[k]) for k in out)
    monkeypatch.setattr(L, "call", real_call)
    eng.close()


def test_forward_pipeline_equals_lone_engine(gpu):
    """ForwardPipeline: three replicas (own stream and activations), frames handed out round-robin: every frame's result is
    the lone engine's, bit for bit and in submission order; the replicas take over the first one's tile plan."""
    from fcn_object_detector_amd.engine import ForwardPipeline
    msg, params, lone = build(1, 96, 128, 3, autotune=False)
    pipe = ForwardPipeline(lambda: NetSpec(msg, "TEST"), params=params, device=0, depth=3, autotune=False)
    rng = np.random.default_rng(9)
    frames = [rng.random((1, 3, 96, 128), dtype=np.float32) for _ in range(8)]
    want = []
    for f in frames:
        lone.host_array("data")[...] = f
        want.append({k: v.copy() for k, v in lone.forward().items()})
    got = pipe.map([{"data": f} for f in frames])
    assert len(got) == 8
    for g, w in zip(got, want):
        for k in ("coverage", "bboxes"):
            assert np.array_equal(g[k], w[k])
    rb = oracle_forward(msg, params, frames[5])
    assert rel_err(got[5]["coverage"], rb["coverage"]) < TOL and rel_err(got[5]["bboxes"], rb["bboxes"]) < TOL
    # flow control: no more than `depth` frames outstanding, nothing to collect when idle
    for f in frames[:3]:
        pipe.submit({"data": f})
    with pytest.raises(RuntimeError):
        pipe.submit({"data": frames[3]})
    assert np.array_equal(pipe.collect()["coverage"], want[0]["coverage"])
    pipe.collect(), pipe.collect()
    with pytest.raises(RuntimeError):
        pipe.collect()
    assert pipe.run_resident(7) > 0
    # a tuned first replica hands its plan to the others
    tuned = ForwardPipeline(lambda: NetSpec(msg, "TEST"), params=params, device=0, depth=2)
    assert tuned.engines[0]._chosen_cfgs and tuned.engines[1]._chosen_cfgs == tuned.engines[0]._chosen_cfgs
    tuned.close()
    pipe.close()
    lone.close()


def test_frames_in_flight_are_reproducible(gpu):
    """2000 copies of one frame through four replicas that share the GPU: every result carries the first one's bits (kernels
    that overlap must not disturb each other - the check that pins the convolution kernel's tail-prefetch fix)."""
    from fcn_object_detector_amd.engine import ForwardPipeline
    msg = proto.parse_text(models.googlenet_detectnet_deploy(1, 96, 128, 4))
    spec = NetSpec(msg, "TEST")
    spec.infer()
    pipe = ForwardPipeline(lambda: NetSpec(msg, "TEST"), params=fill_params(spec, seed=3), device=0, depth=4)
    x = np.random.default_rng(0).random((1, 3, 96, 128), dtype=np.float32)
    outs = pipe.map([{"data": x}] * 2000)
    bad = [i for i, o in enumerate(outs) if not all(np.array_equal(o[k], outs[0][k]) for k in outs[0])]
    assert not bad, bad[:10]
    pipe.close()


def test_pycaffe_front_end(gpu, tmp_path):
    """The reference's calling sequence (fcn_object_detector.py:68-69,82,87,317-328) against our `caffe` package."""
    if PYCAFFE not in sys.path:
        sys.path.insert(0, PYCAFFE)
    import caffe
    proto_path = str(tmp_path / "deploy.prototxt")
    with open(proto_path, "w") as f:
        f.write(models.googlenet_detectnet_deploy(1, 96, 128, 2))
    msg = proto.parse_file(proto_path)
    spec = NetSpec(msg, "TEST")
    spec.infer()
    params = fill_params(spec, seed=7)
    weights = str(tmp_path / "snap.caffemodel")
    proto.write_caffemodel(weights, [(l.name, l.type, params[l.name]) for l in spec.param_layers()])
    with pytest.raises(IOError):
        caffe.Net(proto_path, str(tmp_path / "missing.caffemodel"), caffe.TEST)
    caffe.set_device(0)
    caffe.set_mode_gpu()
    net = caffe.Net(proto_path, weights, caffe.TEST)
    tr = caffe.io.Transformer({"data": net.blobs["data"].data.shape})
    tr.set_transpose("data", (2, 0, 1)); tr.set_raw_scale("data", 1); tr.set_channel_swap("data", (2, 1, 0))
    shape = net.blobs["data"].data.shape
    net.blobs["data"].reshape(1, 3, shape[2], shape[3])
    img = np.random.default_rng(5).random((3, 96, 128))              # float64, like the node's cv_img
    net.blobs["data"].data[0][...] = img
    out = net.forward()
    rb = oracle_forward(msg, params, img[None].astype(np.float32))
    assert set(out) == {"coverage", "bboxes"}
    assert rel_err(net.blobs["coverage"].data[0], rb["coverage"][0]) < TOL
    assert rel_err(net.blobs["bboxes"].data[0], rb["bboxes"][0]) < TOL
    assert rel_err(net.blobs["inception_4c/output"].data, rb["inception_4c/output"]) < TOL
    # run from another thread, as the rospy subscriber callback does
    import threading
    res = {}

    def worker():
        caffe.set_device(0); caffe.set_mode_gpu()
        res["o"] = {k: v.copy() for k, v in net.forward().items()}
    th = threading.Thread(target=worker); th.start(); th.join()
    assert np.array_equal(res["o"]["coverage"], out["coverage"])
    # parameter edit through net.params is visible to the next forward
    net.params["bbox/regressor"][1].data[...] = 3.0
    out2 = net.forward()
    assert np.allclose(out2["bboxes"], rb["bboxes"] + 3.0, rtol=1e-3, atol=1e-3)
    net.save(str(tmp_path / "resaved.caffemodel"))
    assert np.array_equal(proto.read_caffemodel(str(tmp_path / "resaved.caffemodel"))["bbox/regressor"][1], np.full(8, 3.0, np.float32))
