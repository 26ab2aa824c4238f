"""Host logic: prototxt reader, shape rules, fillers, caffemodel codec (CPU only)."""
import os

import numpy as np
import pytest

from fcn_object_detector_amd import models, proto
from fcn_object_detector_amd.netspec import NetSpec, bilinear_kernel, fill_params, pool_out

REF = "/root/reference"


def test_text_format_features():
    txt = """
    name: "n"  # comment
    input: "data" input_dim: 1 input_dim: 3 input_dim: 8 input_dim: 8
    layer { name: 'a' type: 'Python' top: "x" python_param { module: 'm' layer: 'L' param_str : "1,2,/p/train.txt" }
            include: { phase: TRAIN } }
    layer { name: "b" type: "Pooling" bottom: "data" top: "p" pooling_param { pool: MAX kernel_size: 3 stride: 2 } }
    base_lr: 1e-10 flag: true
    """
    m = proto.parse_text(txt)
    assert m.get("name") == "n" and m.get("base_lr") == 1e-10 and m.get("flag") is True
    assert [int(d) for d in m.getall("input_dim")] == [1, 3, 8, 8]
    la, lb = m.getall("layer")
    assert la.get("python_param").get("param_str") == "1,2,/p/train.txt"
    assert la.get("include").get("phase") == "TRAIN"
    assert lb.get("pooling_param").get("pool") == "MAX"
    assert len(NetSpec(m, "TEST").layers) == 1 and len(NetSpec(m, "TRAIN").layers) == 2
    with pytest.raises(ValueError):
        proto.parse_text("layer { name: 'x' ")


def test_pool_shape_rule():
    # ceil mode: 224 -> 112 -> 56 -> 28 for k3 s2 p0 ; k3 s1 p1 keeps the size ; pad clip case
    assert [pool_out(h, 3, 2, 0) for h in (224, 112, 56)] == [112, 56, 28]
    assert pool_out(56, 3, 1, 1) == 56
    assert pool_out(5, 2, 2, 1) == 3          # (5+2-2)/2 = 2.5 -> 3 (+1 = 4) -> clip since 3*2 >= 5+1 -> 3
    assert pool_out(28, 2, 2, 0) == 14


def test_deploy_shapes_and_param_count():
    spec = NetSpec(proto.parse_text(models.googlenet_detectnet_deploy()), "TEST")
    shapes = spec.infer()
    assert shapes["coverage"] == (1, 4, 28, 28) and shapes["bboxes"] == (1, 16, 28, 28)
    assert shapes["conv1/7x7_s2"] == (1, 64, 224, 224) and shapes["inception_5b/output"] == (1, 1024, 28, 28)
    nparam = sum(int(np.prod(s)) for ss in spec.param_shapes.values() for s in ss)
    assert nparam == 5994052
    flops = 0
    for l in spec.layers:
        if l.type == "Convolution":
            co, ci, k, _ = spec.param_shapes[l.name][0]
            n, _, oh, ow = shapes[l.tops[0]]
            flops += 2 * co * ci * k * k * oh * ow
    assert abs(flops / 1e9 - 15.608) < 0.01                    # BASELINE.md forward GFLOP per frame
    assert spec.output_blobs() == ["coverage", "bboxes"]


def test_fillers():
    spec = NetSpec(proto.parse_text(models.googlenet_detectnet_deploy()), "TEST")
    spec.infer()
    p = fill_params(spec, seed=1234)
    w, b = p["conv2/3x3"]
    bound = np.sqrt(3.0 / (64 * 9))
    assert w.shape == (192, 64, 3, 3) and np.abs(w).max() <= bound and np.abs(w).max() > 0.9 * bound
    assert np.all(b == np.float32(0.2)) and np.all(p["bbox/regressor"][1] == 0)
    p2 = fill_params(spec, seed=1234)
    assert all(np.array_equal(a, c) for k in p for a, c in zip(p[k], p2[k]))
    # Caffe BilinearFiller known answers: k=4 -> [0.25 0.75 0.75 0.25] outer product
    k4 = bilinear_kernel(4)
    assert np.allclose(k4[0], [0.0625, 0.1875, 0.1875, 0.0625]) and np.allclose(k4[1, 1], 0.5625)
    assert np.allclose(bilinear_kernel(3)[1], [0.1875, 0.5625, 0.5625])   # f=2, c=0.75 -> [0.25 0.75 0.75] outer


def test_caffemodel_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    layers = [("conv1/7x7_s2", "Convolution", [rng.standard_normal((4, 3, 7, 7)).astype(np.float32),
                                               rng.standard_normal(4).astype(np.float32)]),
              ("up", "Deconvolution", [rng.standard_normal((2, 1, 4, 4)).astype(np.float32)])]
    path = str(tmp_path / "w.caffemodel")
    proto.write_caffemodel(path, layers, "net")
    back = proto.read_caffemodel(path)
    assert list(back) == ["conv1/7x7_s2", "up"]
    for name, _, blobs in layers:
        for a, b in zip(blobs, back[name]):
            assert a.shape == b.shape and np.array_equal(a, b)
    it, hist = proto.unpack_solverstate(proto.pack_solverstate(7, [layers[0][2][1]]))
    assert it == 7 and np.array_equal(hist[0], layers[0][2][1])


def _signature(msg, phase):
    spec = NetSpec(msg, phase)
    out = []
    for l in spec.layers:
        d = [l.type, l.name, tuple(l.bottoms), tuple(l.tops), tuple(l.lr_mult), tuple(l.decay_mult), tuple(l.loss_weight)]
        for key in ("convolution_param", "pooling_param", "lrn_param", "power_param", "dropout_param", "eltwise_param"):
            p = l.msg.get(key)
            if p is None:
                continue
            items = []
            for k, v in sorted(p.fields.items()):
                vals = []
                for x in v:
                    if isinstance(x, proto.Msg):
                        vals.append(tuple((kk, tuple(vv)) for kk, vv in sorted(x.fields.items()) if kk != "std"))
                    else:
                        vals.append(float(x) if isinstance(x, (int, float)) and not isinstance(x, bool) else x)
                items.append((k, tuple(vals)))
            d.append((key, tuple(items)))
        out.append(tuple(d))
    return spec.input_shapes, out


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted (GPU box)")
def test_builder_equals_reference_deploy():
    """models.googlenet_detectnet_deploy() must parse to the same graph as the reference's models/deploy.prototxt."""
    a = _signature(proto.parse_file(os.path.join(REF, "models/deploy.prototxt")), "TEST")
    b = _signature(proto.parse_text(models.googlenet_detectnet_deploy()), "TEST")
    assert a[0] == b[0]
    assert len(a[1]) == len(b[1]) == 142
    for x, y in zip(a[1], b[1]):
        assert x == y


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted (GPU box)")
def test_builder_equals_reference_fcn_bbox_train():
    """models.vgg16_fcn_bbox_train() must parse to the same graph as the reference's train/fcn_bbox/train_val.prototxt."""
    ref = proto.parse_file(os.path.join(REF, "train/fcn_bbox/train_val.prototxt"))
    param_str = [l for l in ref.getall("layer") if l.get("type") == "Python"][0].get("python_param").get("param_str")
    a = _signature(ref, "TRAIN")
    b = _signature(proto.parse_text(models.vgg16_fcn_bbox_train("data_argumentation_layer", "DataArgumentationLayer", param_str)), "TRAIN")
    assert len(a[1]) == len(b[1]) == 50
    for x, y in zip(a[1], b[1]):
        assert x == y


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted (GPU box)")
def test_builder_equals_reference_bounding_box_train():
    ref = proto.parse_file(os.path.join(REF, "train/bounding_box/train_val.prototxt"))
    param_str = [l for l in ref.getall("layer") if l.get("type") == "Python"][0].get("python_param").get("param_str")
    a = _signature(ref, "TRAIN")
    b = _signature(proto.parse_text(models.vgg16_bounding_box_train("data_argumentation_layer", "DataArgumentationLayer", param_str)), "TRAIN")
    assert len(a[1]) == len(b[1]) == 42
    for x, y in zip(a[1], b[1]):
        assert x == y


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted (GPU box)")
def test_builder_equals_reference_models_train_val():
    """The LMDB-fronted training nets as shipped: models/train_val.prototxt and models/train_val2.prototxt (3-class heads)."""
    for rel, kw in (("models/train_val.prototxt", {}), ("models/train_val2.prototxt", {"head_classes": 3})):
        a = _signature(proto.parse_file(os.path.join(REF, rel)), "TRAIN")
        b = _signature(proto.parse_text(models.googlenet_detectnet_train_lmdb(**kw)), "TRAIN")
        assert len(a[1]) == len(b[1]) == 152
        for x, y in zip(a[1], b[1]):
            assert x == y


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted (GPU box)")
def test_builder_equals_reference_bounding_box_deploy():
    a = _signature(proto.parse_file(os.path.join(REF, "train/bounding_box/deploy.prototxt")), "TEST")
    b = _signature(proto.parse_text(models.vgg16_bounding_box_deploy()), "TEST")
    assert a[0] == b[0] == {"data": (10, 3, 448, 448)} and len(a[1]) == len(b[1]) == 51
    for x, y in zip(a[1], b[1]):
        assert x == y


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted (GPU box)")
def test_all_reference_prototxts_parse():
    for rel in ("models/deploy.prototxt", "models/train_val.prototxt", "models/train_val2.prototxt",
                "train/fcn_bbox/train_val.prototxt", "train/bounding_box/train_val.prototxt",
                "train/bounding_box/deploy.prototxt", "train/semantic_segmentation/train_val.prototxt",
                "train/fcn_bbox/solver.prototxt", "train/bounding_box/solver.prototxt"):
        m = proto.parse_file(os.path.join(REF, rel))
        assert m.fields
    s = proto.parse_file(os.path.join(REF, "train/bounding_box/solver.prototxt"))
    assert s.get("solver_type") == "ADAM" and s.get("base_lr") == 1e-4 and s.get("stepsize") == 10000
    tv = NetSpec(proto.parse_file(os.path.join(REF, "train/fcn_bbox/train_val.prototxt")), "TRAIN")
    shapes = tv.infer({"data": (2, 3, 288, 288), "label": (2, 1, 288, 288), "bbox-label": (2, 44, 36, 36),
                       "size-block": (2, 44, 36, 36), "obj-block": (2, 44, 36, 36), "coverage-block": (2, 44, 36, 36)})
    assert shapes["upscore_pool5_bbox"] == (2, 44, 36, 36) and shapes["upscore_pool3"] == (2, 11, 288, 288)
