"""Training-scene synthesis on the device (csrc/scene.hip, DeviceRenderer) against the host renderer (-m gpu)."""
import random

import numpy as np
import pytest

from conftest import rel_err
from fcn_object_detector_amd import data_layer as D
from fcn_object_detector_amd import proto
from fcn_object_detector_amd.engine import Engine
from fcn_object_detector_amd.netspec import NetSpec
from fcn_object_detector_amd.pylayer import TopProxy
from oracle import scene_ref as S

pytestmark = pytest.mark.gpu

NET = """
input: "data"
input_shape { dim: %d dim: 3 dim: %d dim: %d }
input: "label"
input_shape { dim: %d dim: 1 dim: %d dim: %d }
layer { name: "p" type: "Pooling" bottom: "data" top: "p" pooling_param { pool: MAX kernel_size: 2 stride: 2 } }
layer { name: "q" type: "Pooling" bottom: "label" top: "q" pooling_param { pool: MAX kernel_size: 2 stride: 2 } }
"""


def make(mode, n=3, W=160, H=96, classes=4, objects=3):
    lay = D.DataArgumentationLayer()
    lay.param_str = "%d,%d,16,%d,%d,synthetic:%d%s" % (W, H, classes, n, objects, ",detectnet" if mode == "detectnet" else "")
    tops = [TopProxy(t) for t in ("data", "label", "bbox-label", "size-block", "obj-block", "coverage-block")]
    lay.setup([], tops)
    lay.reshape([], tops)
    spec = NetSpec(proto.parse_text(NET % (n, H, W, n, H, W)), "TEST")
    eng = Engine(spec, params={}, device=0, autotune=False)
    lay.bind_device(eng, ["data", "label"])
    return lay, eng, tops


def test_device_scene_equals_host_scene_bit_for_bit(gpu):
    lay, eng, _ = make("mask")
    random.seed(2024)
    seen_flip, seen_scale, seen_clip = set(), 0, 0
    for it in range(40):
        plan = lay.plan_scene()
        lay._renderer.render(it % 3, plan)
        img_d, msk_d = lay._renderer.read_scene(it % 3)
        img_h, msk_h = S.render_scene(lay, plan)
        assert np.array_equal(img_d, img_h), it
        assert np.array_equal(msk_d, msk_h), it
        seen_flip.add(plan["final_flip"])
        for o in plan["objects"]:
            seen_flip.add(("obj", o["flip"]))
            seen_scale += o["out"] != o["roi"][2:]
            seen_clip += o["pos"][0] < 0 or o["pos"][1] < 0 or o["pos"][0] + o["out"][0] > 640 or o["pos"][1] + o["out"][1] > 480
        assert set(np.unique(msk_h)) <= {0, 1, 2, 3}
    assert {("obj", -1), ("obj", 0), ("obj", 1), ("obj", 2)} <= seen_flip and seen_scale > 5      # the cases were really exercised
    eng.close()


@pytest.mark.parametrize("mode", ["mask", "detectnet"])
def test_device_fed_tops_match_host_tops(gpu, mode):
    """`data` from compose + fcn_preprocess_bgr8 vs the host chain (demean f32, min-max, bilinear resize); the class mask of
    HEAD's mode vs the nearest-neighbour resize; rects / labels identical (same plan).  The layer has no host renderer:
    without a bound engine forward() fails loudly."""
    lay, eng, tops = make(mode)
    unbound = D.DataArgumentationLayer()
    unbound.param_str = lay.param_str
    unbound.setup([], tops)
    with pytest.raises(RuntimeError):
        unbound.forward([], tops)
    random.seed(7)
    state = random.getstate()
    lay.forward([], tops)                       # device path: renders into the engine's blobs
    dev_rects, dev_labels = lay.last_rects, lay.last_labels
    eng.blobs["data"].host_valid = False          # written in HBM by the renderer, not through the host array
    data_dev = eng.read_blob("data").copy()
    random.setstate(state)
    host = np.zeros((lay.batch_size, 3, lay.image_size_y, lay.image_size_x), np.float32)
    host_mask = np.zeros((lay.batch_size, 1, lay.image_size_y, lay.image_size_x), np.float32)
    host_rects, host_labels = [], []
    for i in range(lay.batch_size):             # the oracle's renderer with the same draws
        img, mask, rects, labels = S.make_sample(lay)
        host[i] = img.transpose(2, 0, 1)
        host_mask[i, 0] = mask
        host_rects.append(rects)
        host_labels.append(labels)
    assert host_rects == dev_rects and host_labels == dev_labels
    assert rel_err(data_dev, host) < 1e-5
    assert np.abs(data_dev - host).max() < 2e-6
    if mode == "mask":
        eng.blobs["label"].host_valid = False
        assert np.array_equal(eng.read_blob("label"), host_mask)
    eng.close()
