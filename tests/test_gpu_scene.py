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
    lay._color_rng = np.random.default_rng(2024)
    seen_flip, seen_scale, seen_clip, seen_zoom, seen_blur = set(), 0, 0, 0, set()
    for it in range(60):
        plan = lay.plan_scene()
        lay._renderer.render(it % 3, plan)
        img_d, msk_d = lay._renderer.read_scene(it % 3)
        img_h, msk_h = S.render_scene(lay, plan)
        assert img_d.shape == img_h.shape
        assert np.array_equal(img_d, img_h), (it, plan["view"], plan["color"])
        assert np.array_equal(msk_d, msk_h), it
        seen_flip.add(plan["final_flip"])
        seen_zoom += plan["view"] is not None
        seen_blur.add(plan["color"]["blur"]["kind"])
        for o in plan["objects"]:
            seen_flip.add(("obj", o["flip"]))
            seen_scale += o["out"] != o["roi"][2:]
            seen_clip += o["pos"][0] < 0 or o["pos"][1] < 0 or o["pos"][0] + o["out"][0] > 640 or o["pos"][1] + o["out"][1] > 480
        assert set(np.unique(msk_h)) <= {0, 1, 2, 3}
    assert {("obj", -1), ("obj", 0), ("obj", 1), ("obj", 2)} <= seen_flip and seen_scale > 5      # the cases were really exercised
    assert seen_zoom >= 5 and seen_blur == {"gauss", "box", "median"}
    eng.close()


def _dev_image_op(name, img, *args):
    import ctypes as C
    from fcn_object_detector_amd import lib as L
    from gpu_util import dev_from, dev_to
    src, dst = dev_from(img), dev_from(np.zeros_like(img))
    L.call(name, src.ptr, dst.ptr, *args, None)
    return dev_to(dst, img.shape, np.uint8)


@pytest.mark.parametrize("hw", [(1, 1), (2, 9), (7, 5), (33, 41), (96, 130)])
def test_colour_kernels_match_oracle(gpu, hw):
    """Each operator of the restated imgaug sequence on its own, bit for bit, down to images smaller than the kernels
    (the reflect-101 border then folds more than once)."""
    import ctypes as C
    from fcn_object_detector_amd import lib as L
    from gpu_util import dev_from
    h, w = hw
    rng = np.random.default_rng(h * 100 + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    tmp = dev_from(np.zeros((h, w, 3), np.float32))
    for sigma in (0.05, 0.7, 1.9, 3.0):
        taps = D.gauss_taps(sigma)
        got = _dev_image_op("fcn_blur_gauss_bgr8", img, tmp.ptr, h, w, taps.ctypes.data, len(taps) - 1)
        assert np.array_equal(got, S.blur_gauss(img, taps)), sigma
    for k in range(1, 9):
        assert np.array_equal(_dev_image_op("fcn_blur_box_bgr8", img, h, w, k), S.blur_box(img, k)), k
    for k in (3, 5, 7):
        assert np.array_equal(_dev_image_op("fcn_blur_median_bgr8", img, h, w, k), S.blur_median(img, k)), k
    for _ in range(6):
        c = D.plan_color(rng)
        al, light = c["sharpen"]
        ga = np.float32(c["gray"])
        q = L.ColorParams(float(np.float32((1.0 - al) + al * (8.0 + light))), float(np.float32(-al)), (C.c_int32 * 3)(*c["add"]),
                          (C.c_float * 3)(*c["mul"]), float(ga), float(np.float32(1.0) - ga))
        got = _dev_image_op("fcn_color_augment_bgr8", img, h, w, C.byref(q))
        assert np.array_equal(got, S.color_point_ops(img, c["sharpen"], c["add"], c["mul"], c["gray"]))
    lib = L.load()
    src = dev_from(img)
    assert lib.fcn_blur_median_bgr8(src.ptr, tmp.ptr, h, w, 4, None) != 0 and lib.fcn_blur_box_bgr8(src.ptr, tmp.ptr, h, w, 16, None) != 0
    assert lib.fcn_blur_box_bgr8(src.ptr, src.ptr, h, w, 3, None) != 0                 # in place is refused


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
    lay._color_rng = np.random.default_rng(7)
    state, cstate = random.getstate(), lay._color_rng.bit_generator.state
    lay.forward([], tops)                       # device path: renders into the engine's blobs
    dev_rects, dev_labels = lay.last_rects, lay.last_labels
    eng.blobs["data"].host_valid = False          # written in HBM by the renderer, not through the host array
    data_dev = eng.read_blob("data").copy()
    random.setstate(state)
    lay._color_rng.bit_generator.state = cstate
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
