"""Programmatic builders for the detector networks the reference ships as prototxt.

The engine consumes the reference's prototxt files unmodified (``caffe.Net(path, ...)``).  These
builders exist because /root/reference is not available on the GPU box: bench.py, smoke() and the
GPU tests need the same networks without carrying the reference's files.  They emit prototxt TEXT
that parses to the same layer graph (names, types, bottoms/tops, kernel geometry, fillers,
lr/decay multipliers) as

  * ``googlenet_detectnet_deploy``  <->  reference models/deploy.prototxt
  * ``googlenet_detectnet_train``   <->  reference models/train_val.prototxt with the LMDB Data +
                                         Slice front replaced by the Python data layer's tops, as the
                                         reference README (:57-76) instructs
  * ``vgg16_fcn_bbox``              <->  reference train/fcn_bbox/train_val.prototxt (bbox branch + seg branch)

tests/test_models.py checks that equivalence layer by layer whenever /root/reference is present.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

# (module, 1x1, 3x3_reduce, 3x3, 5x5_reduce, 5x5, pool_proj) — GoogLeNet v1 widths
INCEPTION = [
    ("inception_3a", 64, 96, 128, 16, 32, 32),
    ("inception_3b", 128, 128, 192, 32, 96, 64),
    ("inception_4a", 192, 96, 208, 16, 48, 64),
    ("inception_4b", 160, 112, 224, 24, 64, 64),
    ("inception_4c", 128, 128, 256, 24, 64, 64),
    ("inception_4d", 112, 144, 288, 32, 64, 64),
    ("inception_4e", 256, 160, 320, 32, 128, 128),
    ("inception_5a", 256, 160, 320, 32, 128, 128),
    ("inception_5b", 384, 192, 384, 48, 128, 128),
]


# the reference gives this one bias lr_mult 1 instead of 2 (models/deploy.prototxt:1926-1957); reproduced, not fixed
BIAS_LR_QUIRK = {"inception_5b/3x3_reduce": (1.0, 1.0)}


class _Writer:
    def __init__(self) -> None:
        self.lines: List[str] = []

    def raw(self, s: str) -> None:
        self.lines.append(s)

    def layer(self, name: str, type_: str, bottoms: Sequence[str], tops: Sequence[str], body: str = "",
              quote: str = '"', extra: str = "") -> None:
        s = ["layer {", '  name: "%s"' % name, "  type: %s%s%s" % (quote, type_, quote)]
        s += ['  bottom: "%s"' % b for b in bottoms]
        s += ['  top: "%s"' % t for t in tops]
        if extra:
            s.append(extra)
        if body:
            s.append(body)
        s.append("}")
        self.lines.append("\n".join(s))

    def text(self) -> str:
        return "\n".join(self.lines) + "\n"


def _conv_body(num_output: int, k: int, pad: int = 0, stride: int = 1, bias_value: float = 0.2,
               lr: Tuple[float, float] = (1.0, 2.0), decay: Tuple[float, float] = (1.0, 0.0), explicit: bool = False) -> str:
    s = ["  param { lr_mult: %g decay_mult: %g }" % (lr[0], decay[0]),
         "  param { lr_mult: %g decay_mult: %g }" % (lr[1], decay[1]),
         "  convolution_param {", "    num_output: %d" % num_output]
    if pad or explicit:                      # explicit: True = write pad and stride even at their defaults, "pad" = pad only
        s.append("    pad: %d" % pad)
    s.append("    kernel_size: %d" % k)
    if stride != 1 or explicit is True:
        s.append("    stride: %d" % stride)
    s += ['    weight_filler { type: "xavier" }', '    bias_filler { type: "constant" value: %g }' % bias_value, "  }"]
    return "\n".join(s)


def _conv_relu(w: _Writer, name: str, relu_name: str, bottom: str, num_output: int, k: int, pad: int = 0, stride: int = 1) -> None:
    w.layer(name, "Convolution", [bottom], [name], _conv_body(num_output, k, pad, stride, lr=BIAS_LR_QUIRK.get(name, (1.0, 2.0))))
    w.layer(relu_name, "ReLU", [name], [name])


def _pool(w: _Writer, name: str, bottom: str, k: int, stride: int, pad: int = 0) -> None:
    body = "  pooling_param { pool: MAX kernel_size: %d stride: %d%s }" % (k, stride, " pad: %d" % pad if pad else "")
    w.layer(name, "Pooling", [bottom], [name], body)


def _lrn(w: _Writer, name: str, bottom: str) -> None:
    w.layer(name, "LRN", [bottom], [name], "  lrn_param { local_size: 5 alpha: 0.0001 beta: 0.75 }")


def _googlenet_body(w: _Writer, data_blob: str) -> str:
    w.layer("deploy_transform", "Power", [data_blob], ["transformed_data"], "  power_param { shift: -127.0 }")
    _conv_relu(w, "conv1/7x7_s2", "conv1/relu_7x7", "transformed_data", 64, 7, 3, 2)
    _pool(w, "pool1/3x3_s2", "conv1/7x7_s2", 3, 2)
    _lrn(w, "pool1/norm1", "pool1/3x3_s2")
    _conv_relu(w, "conv2/3x3_reduce", "conv2/relu_3x3_reduce", "pool1/norm1", 64, 1)
    _conv_relu(w, "conv2/3x3", "conv2/relu_3x3", "conv2/3x3_reduce", 192, 3, 1)
    _lrn(w, "conv2/norm2", "conv2/3x3")
    _pool(w, "pool2/3x3_s2", "conv2/norm2", 3, 2)
    prev = "pool2/3x3_s2"
    for mod, c1, c3r, c3, c5r, c5, cp in INCEPTION:
        _conv_relu(w, mod + "/1x1", mod + "/relu_1x1", prev, c1, 1)
        _conv_relu(w, mod + "/3x3_reduce", mod + "/relu_3x3_reduce", prev, c3r, 1)
        _conv_relu(w, mod + "/3x3", mod + "/relu_3x3", mod + "/3x3_reduce", c3, 3, 1)
        _conv_relu(w, mod + "/5x5_reduce", mod + "/relu_5x5_reduce", prev, c5r, 1)
        _conv_relu(w, mod + "/5x5", mod + "/relu_5x5", mod + "/5x5_reduce", c5, 5, 2)
        _pool(w, mod + "/pool", prev, 3, 1, 1)
        _conv_relu(w, mod + "/pool_proj", mod + "/relu_pool_proj", mod + "/pool", cp, 1)
        w.layer(mod + "/output", "Concat", [mod + "/1x1", mod + "/3x3", mod + "/5x5", mod + "/pool_proj"], [mod + "/output"])
        prev = mod + "/output"
        if mod == "inception_3b":
            _pool(w, "pool3/3x3_s2", prev, 3, 2)
            prev = "pool3/3x3_s2"
    w.layer("pool5/drop_s1", "Dropout", [prev], ["pool5/drop_s1"], "  dropout_param { dropout_ratio: 0.4 }")
    return "pool5/drop_s1"


def _heads(w: _Writer, feat: str, num_classes: int) -> None:
    w.layer("cvg/classifier", "Convolution", [feat], ["cvg/classifier"], _conv_body(num_classes, 1, bias_value=0.0))
    w.layer("coverage/sig", "Sigmoid", ["cvg/classifier"], ["coverage"])
    w.layer("bbox/regressor", "Convolution", [feat], ["bboxes"], _conv_body(4 * num_classes, 1, bias_value=0.0))


def googlenet_detectnet_deploy(batch: int = 1, height: int = 448, width: int = 448, num_classes: int = 4) -> str:
    """Inference net: input ``data`` -> ``coverage`` (C x H/16 x W/16) and ``bboxes`` (4C x H/16 x W/16)."""
    w = _Writer()
    w.raw('input: "data"\ninput_shape {\n  dim: %d\n  dim: 3\n  dim: %d\n  dim: %d\n}' % (batch, height, width))
    feat = _googlenet_body(w, "data")
    _heads(w, feat, num_classes)
    return w.text()


def googlenet_detectnet_train(module: str, layer: str, param_str: str, num_classes: int = 1) -> str:
    """Training net: Python data layer (6 tops) -> GoogLeNet body -> masked/normalised L1 + Euclidean losses."""
    w = _Writer()
    tops = ["data", "coverage-label", "bbox-label", "size-block", "obj-block", "coverage-block"]
    body = "  python_param {\n    module: '%s'\n    layer: '%s'\n    param_str: '%s'\n  }" % (module, layer, param_str)
    w.layer("data", "Python", [], tops, body, quote="'")
    prod = "  eltwise_param { operation: PROD }"
    w.layer("bb-label-norm", "Eltwise", ["bbox-label", "size-block"], ["bbox-label-norm"], prod)
    w.layer("bb-obj-norm", "Eltwise", ["bbox-label-norm", "obj-block"], ["bbox-obj-label-norm"], prod)
    feat = _googlenet_body(w, "data")
    _heads(w, feat, num_classes)
    w.layer("bbox_mask", "Eltwise", ["bboxes", "coverage-block"], ["bboxes-masked"], prod)
    w.layer("bbox-norm", "Eltwise", ["bboxes-masked", "size-block"], ["bboxes-masked-norm"], prod)
    w.layer("bbox-obj-norm", "Eltwise", ["bboxes-masked-norm", "obj-block"], ["bboxes-obj-masked-norm"], prod)
    w.layer("bbox_loss", "L1Loss", ["bboxes-obj-masked-norm", "bbox-obj-label-norm"], ["loss_bbox"], extra="  loss_weight: 2.0")
    w.layer("coverage_loss", "EuclideanLoss", ["coverage", "coverage-label"], ["loss_coverage"])
    return w.text()


def googlenet_detectnet_train_lmdb(features_db: str = "/home/krishneel/Desktop/lmdb/features", labels_db: str = "/home/krishneel/Desktop/lmdb/labels",
                                   batch: int = 1, num_classes: int = 1, head_classes: Optional[int] = None) -> str:
    """The reference's models/train_val.prototxt as it stands (models/train_val2.prototxt = head_classes 3 over the SAME
    1-class slice points: the reference's own inconsistency, reproduced when asked for): two LMDB `Data` layers (image, 17-channel label record), a
    `Slice` that cuts the record into coverage-label / bbox-label / size-block / obj-block / coverage-block, then the same
    body and loss tail as googlenet_detectnet_train.  LMDB reading is out of scope: with this engine the two
    Data tops are input blobs the caller fills (pycaffe `net.blobs['data'].data[...] = ...`)."""
    w = _Writer()
    for name, top, src in (("train_data", "data", features_db), ("train_label", "label", labels_db)):
        body = "  include { phase: TRAIN }\n  data_param {\n    source: \"%s\"\n    batch_size: %d\n    backend: LMDB\n  }" % (src, batch)
        w.layer(name, "Data", [], [top], body)
    c = num_classes
    pts = (c, 5 * c, 9 * c, 13 * c)
    slice_tops = ["coverage-label", "bbox-label", "size-block", "obj-block", "coverage-block"]
    slice_body = "  slice_param {\n    slice_dim: 1\n" + "".join("    slice_point: %d\n" % p for p in pts) + "  }"
    prod = "  eltwise_param { operation: PROD }"
    feat_w = _Writer()
    feat = _googlenet_body(feat_w, "data")
    body_layers = feat_w.lines
    w.lines.append(body_layers[0])                                   # deploy_transform (Power) sits in front of the Slice
    w.layer("slice-label", "Slice", ["label"], slice_tops, slice_body)
    w.layer("bb-label-norm", "Eltwise", ["bbox-label", "size-block"], ["bbox-label-norm"], prod)
    w.layer("bb-obj-norm", "Eltwise", ["bbox-label-norm", "obj-block"], ["bbox-obj-label-norm"], prod)
    w.lines.extend(body_layers[1:])
    _heads(w, feat, head_classes or num_classes)
    w.layer("bbox_mask", "Eltwise", ["bboxes", "coverage-block"], ["bboxes-masked"], prod)
    w.layer("bbox-norm", "Eltwise", ["bboxes-masked", "size-block"], ["bboxes-masked-norm"], prod)
    w.layer("bbox-obj-norm", "Eltwise", ["bboxes-masked-norm", "obj-block"], ["bboxes-obj-masked-norm"], prod)
    w.layer("bbox_loss", "L1Loss", ["bboxes-obj-masked-norm", "bbox-obj-label-norm"], ["loss_bbox"], extra="  loss_weight: 2.0")
    w.layer("coverage_loss", "EuclideanLoss", ["coverage", "coverage-label"], ["loss_coverage"])
    return w.text()


# VGG16 conv stack: (block, number of convs, width)
VGG16 = [(1, 2, 64), (2, 2, 128), (3, 3, 256), (4, 3, 512), (5, 3, 512)]


def _vgg16_body(w: _Writer, data_blob: str) -> None:
    prev = data_blob
    for blk, n, width_ in VGG16:
        for i in range(1, n + 1):
            nm = "conv%d_%d" % (blk, i)
            w.layer(nm, "Convolution", [prev], [nm], _conv_body(width_, 3, 1, explicit=True))
            w.layer("relu%d_%d" % (blk, i), "ReLU", [nm], [nm])
            prev = nm
        body = "  pooling_param { pool: MAX kernel_size: 2 stride: 2 }"
        w.layer("pool%d" % blk, "Pooling", [prev], ["pool%d" % blk], body)
        prev = "pool%d" % blk
    w.layer("dropout5", "Dropout", ["pool5"], ["dropout5"], "  dropout_param { dropout_ratio: 0.5 }")


def _frozen_bilinear_deconv(w: _Writer, name: str, bottom: str, ch: int, k: int, s: int, p: int) -> None:
    body = ("  convolution_param {\n    kernel_size: %d\n    stride: %d\n    num_output: %d\n    group: %d\n    pad: %d\n"
            "    weight_filler { type: \"bilinear\" }\n    bias_term: false\n  }\n  param { lr_mult: 0 decay_mult: 0 }") % (k, s, ch, ch, p)
    w.layer(name, "Deconvolution", [bottom], [name], body)


def _fcn_bbox_heads(w: _Writer, num_classes: int) -> None:
    """bbox branch (stride 8 after a x4 bilinear deconvolution) and the FCN-8s style score branch of train/fcn_bbox."""
    c4 = 4 * num_classes
    w.layer("score_conv5_bbox", "Convolution", ["dropout5"], ["score_conv5_bbox"], _conv_body(c4, 1, explicit=True))
    _frozen_bilinear_deconv(w, "upscore_pool5_bbox", "score_conv5_bbox", c4, 8, 4, 2)


def _fcn_bbox_scores(w: _Writer, num_classes: int) -> None:
    w.layer("score_conv5", "Convolution", ["dropout5"], ["score_conv5"], _conv_body(num_classes, 1, explicit=True))
    _frozen_bilinear_deconv(w, "upscore_pool5", "score_conv5", num_classes, 4, 2, 1)
    w.layer("score_pool4", "Convolution", ["pool4"], ["score_pool4"], _conv_body(num_classes, 1, explicit=True))
    w.layer("fuse_pool4", "Eltwise", ["upscore_pool5", "score_pool4"], ["fuse_pool4"], "  eltwise_param { operation: SUM }")
    _frozen_bilinear_deconv(w, "upscore_pool4", "fuse_pool4", num_classes, 4, 2, 1)
    w.layer("score_pool3", "Convolution", ["pool3"], ["score_pool3"], _conv_body(num_classes, 1, explicit=True))
    w.layer("fuse_pool3", "Eltwise", ["upscore_pool4", "score_pool3"], ["fuse_pool3"], "  eltwise_param { operation: SUM }")
    _frozen_bilinear_deconv(w, "upscore_pool3", "fuse_pool3", num_classes, 16, 8, 4)


def vgg16_fcn_bbox_train(module: str, layer: str, param_str: str, num_classes: int = 11) -> str:
    """The reference's train/fcn_bbox/train_val.prototxt (the net HEAD's Python layer and the ROS node match): VGG16 ->
    masked / normalised L1 loss on the x4-upsampled bbox map (stride 8) + SoftmaxWithLoss on the FCN-8s score map against
    the full-resolution class mask the data layer emits as top[1]."""
    w = _Writer()
    tops = ["data", "label", "bbox-label", "size-block", "obj-block", "coverage-block"]
    body = "  python_param {\n    module: '%s'\n    layer: '%s'\n    param_str: '%s'\n  }" % (module, layer, param_str)
    w.layer("Argumentation", "Python", [], tops, body, quote="'")
    _vgg16_body(w, "data")
    _fcn_bbox_heads(w, num_classes)
    prod = "  eltwise_param { operation: PROD }"
    w.layer("bb-label-norm", "Eltwise", ["bbox-label", "size-block"], ["bbox-label-norm"], prod)
    w.layer("bb-obj-norm", "Eltwise", ["bbox-label-norm", "obj-block"], ["bbox-obj-label-norm"], prod)
    w.layer("bbox_mask", "Eltwise", ["upscore_pool5_bbox", "coverage-block"], ["bboxes-masked"], prod)
    w.layer("bbox-norm", "Eltwise", ["bboxes-masked", "size-block"], ["bboxes-masked-norm"], prod)
    w.layer("bbox-obj-norm", "Eltwise", ["bboxes-masked-norm", "obj-block"], ["bboxes-obj-masked-norm"], prod)
    w.layer("bbox_loss", "L1Loss", ["bboxes-obj-masked-norm", "bbox-obj-label-norm"], ["loss_bbox"], extra="  loss_weight: 2.0")
    _fcn_bbox_scores(w, num_classes)
    w.layer("loss", "SoftmaxWithLoss", ["upscore_pool3", "label"], ["loss"], "  loss_param { normalize: false }")
    return w.text()


def vgg16_fcn_bbox_deploy(batch: int = 1, height: int = 448, width: int = 448, num_classes: int = 11) -> str:
    """Inference form of train/fcn_bbox: the node (scripts/fcn_object_detector.py:89-90) reads ``pool_score`` (class
    probabilities at stride 8: Softmax of ``fuse_pool3``) and ``upscore_pool5_bbox``."""
    w = _Writer()
    w.raw('input: "data"\ninput_shape {\n  dim: %d\n  dim: 3\n  dim: %d\n  dim: %d\n}' % (batch, height, width))
    _vgg16_body(w, "data")
    _fcn_bbox_heads(w, num_classes)
    _fcn_bbox_scores(w, num_classes)
    w.layer("pool_score", "Softmax", ["fuse_pool3"], ["pool_score"])
    return w.text()


def vgg16_bounding_box_train(module: str, layer: str, param_str: str, num_classes: int = 11) -> str:
    """The reference's train/bounding_box/train_val.prototxt (solver: ADAM, step policy): VGG16 with conv1_1..conv3_3
    frozen (lr_mult 0), no ReLU after conv5_3, a frozen x2 bilinear deconvolution back to stride 8, dropout, the
    DetectNet coverage / bbox heads and their L1 + Euclidean losses."""
    w = _Writer()
    tops = ["data", "coverage-label", "bbox-label", "size-block", "obj-block", "coverage-block"]
    body = "  python_param {\n    module: '%s'\n    layer: '%s'\n    param_str: '%s'\n  }" % (module, layer, param_str)
    w.layer("Argumentation", "Python", [], tops, body, quote="'")
    prod = "  eltwise_param { operation: PROD }"
    w.layer("bb-label-norm", "Eltwise", ["bbox-label", "size-block"], ["bbox-label-norm"], prod)
    w.layer("bb-obj-norm", "Eltwise", ["bbox-label-norm", "obj-block"], ["bbox-obj-label-norm"], prod)
    prev = "data"
    for blk, n, width_ in VGG16:
        frozen = blk <= 3
        for i in range(1, n + 1):
            nm = "conv%d_%d" % (blk, i)
            w.layer(nm, "Convolution", [prev], [nm], _conv_body(width_, 3, 1, bias_value=0.0, lr=(0.0, 0.0) if frozen else (1.0, 2.0),
                                                                decay=(0.0, 0.0) if frozen else (1.0, 0.0)))
            if (blk, i) != (5, 3):
                w.layer("relu%d_%d" % (blk, i), "ReLU", [nm], [nm])
            prev = nm
        if blk < 5:
            w.layer("pool%d" % blk, "Pooling", [prev], ["pool%d" % blk], "  pooling_param { pool: MAX kernel_size: 2 stride: 2 }")
            prev = "pool%d" % blk
    _frozen_bilinear_deconv(w, "conv5_3/upsample", "conv5_3", 512, 4, 2, 1)
    w.layer("dropout5", "Dropout", ["conv5_3/upsample"], ["dropout5"], "  dropout_param { dropout_ratio: 0.5 }")
    w.layer("cvg/classifier", "Convolution", ["dropout5"], ["cvg/classifier"], _conv_body(num_classes, 1, bias_value=0.0))
    w.layer("coverage/sig", "Sigmoid", ["cvg/classifier"], ["coverage"])
    w.layer("bbox/regressor", "Convolution", ["dropout5"], ["bboxes"], _conv_body(4 * num_classes, 1, bias_value=0.0))
    w.layer("bbox_mask", "Eltwise", ["bboxes", "coverage-block"], ["bboxes-masked"], prod)
    w.layer("bbox-norm", "Eltwise", ["bboxes-masked", "size-block"], ["bboxes-masked-norm"], prod)
    w.layer("bbox-obj-norm", "Eltwise", ["bboxes-masked-norm", "obj-block"], ["bboxes-obj-masked-norm"], prod)
    w.layer("bbox_loss", "L1Loss", ["bboxes-obj-masked-norm", "bbox-obj-label-norm"], ["loss_bbox"], extra="  loss_weight: 2.0")
    w.layer("coverage_loss", "EuclideanLoss", ["coverage", "coverage-label"], ["loss_coverage"])
    return w.text()


def vgg16_bounding_box_deploy(batch: int = 10, height: int = 448, width: int = 448, num_classes: int = 20) -> str:
    """The reference's train/bounding_box/deploy.prototxt: VGG16 to conv5_3 (stride 16), a pyramid-pooling context branch on
    conv4_3 (AVE pools to 1x1 / 2x2 / 4x4 / 7x7, 1x1 convolutions to 128 channels, frozen bilinear deconvolutions back to
    28x28), concatenated with conv5_3 and pool4 (1536 channels), dropout, the DetectNet coverage / bbox heads."""
    w = _Writer()
    w.layer("data", "Input", [], ["data"], "  input_param { shape { dim: %d dim: 3 dim: %d dim: %d } }" % (batch, height, width))
    prev = "data"
    for blk, n, width_ in VGG16[:4]:
        for i in range(1, n + 1):
            nm = "conv%d_%d" % (blk, i)
            w.layer(nm, "Convolution", [prev], [nm], _conv_body(width_, 3, 1, bias_value=0.0))
            w.layer("relu%d_%d" % (blk, i), "ReLU", [nm], [nm])
            prev = nm
        w.layer("pool%d" % blk, "Pooling", [prev], ["pool%d" % blk], "  pooling_param { pool: MAX kernel_size: 2 stride: 2 }")
        prev = "pool%d" % blk
    ups = []
    for tag, pk, dk, ds, dp in (("1x1", 56, 56, 28, 14), ("2x2", 28, 28, 14, 7), ("4x4", 14, 13, 7, 3), ("7x7", 8, 8, 4, 2)):
        w.layer("pool4/" + tag, "Pooling", ["conv4_3"], ["pool4/" + tag], "  pooling_param { pool: AVE kernel_size: %d stride: %d }" % (pk, pk))
        cn = "conv4_3/" + tag
        w.layer(cn, "Convolution", ["pool4/" + tag], [cn], _conv_body(128, 1, 0, bias_value=0.0, lr=(10.0, 2.0), explicit="pad"))
        w.layer("act_4_3/" + tag, "ReLU", [cn], [cn])
        _frozen_bilinear_deconv(w, cn + "/upsample", cn, 128, dk, ds, dp)
        ups.append(cn + "/upsample")
    prev = "pool4"
    for i in range(1, 4):
        nm = "conv5_%d" % i
        w.layer(nm, "Convolution", [prev], [nm], _conv_body(512, 3, 1, bias_value=0.0))
        if i < 3:
            w.layer("relu5_%d" % i, "ReLU", [nm], [nm])
        prev = nm
    w.layer("conv4_3/conv5_3/concat", "Concat", ["conv5_3", "pool4"] + ups, ["conv4_3/conv5_3/concat"])
    w.layer("dropout5", "Dropout", ["conv4_3/conv5_3/concat"], ["dropout5"], "  dropout_param { dropout_ratio: 0.5 }")
    w.layer("cvg/classifier", "Convolution", ["dropout5"], ["cvg/classifier"], _conv_body(num_classes, 1, bias_value=0.0))
    w.layer("coverage/sig", "Sigmoid", ["cvg/classifier"], ["coverage"])
    w.layer("bbox/regressor", "Convolution", ["dropout5"], ["bboxes"], _conv_body(4 * num_classes, 1, bias_value=0.0))
    return w.text()
