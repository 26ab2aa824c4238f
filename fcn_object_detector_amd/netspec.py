"""Network description: layers, blob shapes and parameter fillers from a Caffe prototxt.

This is the host-side "program" the HIP engine executes — the reference's
models/*.prototxt and train/**/*.prototxt are consumed unmodified (reference:
scripts/fcn_object_detector.py:317 ``caffe.Net(proto, weights, caffe.TEST)``).
Shape rules follow the public Caffe layer definitions (conv: floor, pooling:
ceil with the last-window clip, deconvolution: s(H-1)+k-2p).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import proto

Shape = Tuple[int, ...]


class Layer:
    __slots__ = ("name", "type", "bottoms", "tops", "msg", "lr_mult", "decay_mult", "loss_weight")

    def __init__(self, msg: proto.Msg):
        self.msg = msg
        self.name: str = str(msg.get("name", ""))
        self.type: str = str(msg.get("type", ""))
        self.bottoms: List[str] = [str(b) for b in msg.getall("bottom")]
        self.tops: List[str] = [str(t) for t in msg.getall("top")]
        params = msg.getall("param")
        self.lr_mult = [float(p.get("lr_mult", 1.0)) for p in params]
        self.decay_mult = [float(p.get("decay_mult", 1.0)) for p in params]
        self.loss_weight = [float(w) for w in msg.getall("loss_weight")]

    def sub(self, key: str) -> proto.Msg:
        m = self.msg.get(key)
        return m if m is not None else proto.Msg()

    def __repr__(self) -> str:
        return "Layer(%s:%s %s->%s)" % (self.type, self.name, self.bottoms, self.tops)


def _phase_included(msg: proto.Msg, phase: str) -> bool:
    inc = msg.getall("include")
    if inc:
        return any(str(m.get("phase", phase)) == phase for m in inc)
    exc = msg.getall("exclude")
    if exc:
        return not any(str(m.get("phase", "")) == phase for m in exc)
    return True


def kernel_stride_pad(p: proto.Msg) -> Tuple[int, int, int]:
    k = p.get("kernel_size")
    if k is None:
        k = p.get("kernel_h")
    if k is None:
        raise ValueError("layer without kernel_size")
    if p.get("kernel_w") is not None and int(p.get("kernel_w")) != int(k):
        raise NotImplementedError("non-square kernels are not used by the reference nets")
    return int(k), int(p.get("stride", 1)), int(p.get("pad", 0))


def conv_out(h: int, k: int, s: int, p: int) -> int:
    return (h + 2 * p - k) // s + 1


def pool_out(h: int, k: int, s: int, p: int) -> int:
    o = int(math.ceil((h + 2 * p - k) / float(s))) + 1
    if p > 0 and (o - 1) * s >= h + p:
        o -= 1
    return o


def deconv_out(h: int, k: int, s: int, p: int) -> int:
    return s * (h - 1) + k - 2 * p


DATA_TYPES = ("Data", "Python", "Input", "DummyData", "MemoryData", "ImageData", "HDF5Data")
LOSS_TYPES = ("L1Loss", "EuclideanLoss", "SoftmaxWithLoss", "SigmoidCrossEntropyLoss")


class NetSpec:
    """Phase-filtered layer list + blob/parameter shapes."""

    def __init__(self, msg: proto.Msg, phase: str = "TEST"):
        if phase not in ("TRAIN", "TEST"):
            raise ValueError("phase must be 'TRAIN' or 'TEST'")
        self.phase = phase
        self.name = str(msg.get("name", ""))
        self.layers: List[Layer] = [Layer(m) for m in msg.getall("layer") if _phase_included(m, phase)]
        if msg.getall("layers"):
            raise NotImplementedError("V1 'layers' prototxt syntax is not used by the reference")
        self.input_shapes: Dict[str, Shape] = {}
        names = [str(n) for n in msg.getall("input")]
        shapes = msg.getall("input_shape")
        dims = [int(d) for d in msg.getall("input_dim")]
        for i, nm in enumerate(names):
            if shapes:
                self.input_shapes[nm] = tuple(int(d) for d in shapes[i].getall("dim"))
            else:
                self.input_shapes[nm] = tuple(dims[4 * i:4 * i + 4])
        for l in self.layers:
            if l.type == "Input":
                shp = l.sub("input_param").getall("shape")
                for t, s in zip(l.tops, shp):
                    self.input_shapes[t] = tuple(int(d) for d in s.getall("dim"))
        self.blob_shapes: Dict[str, Shape] = {}
        self.param_shapes: Dict[str, List[Shape]] = {}

    @classmethod
    def from_file(cls, path: str, phase: str = "TEST") -> "NetSpec":
        return cls(proto.parse_file(path), phase)

    # ------------------------------------------------------------------
    def data_tops(self) -> List[str]:
        """Blobs that must be fed from outside: net inputs and tops of data / Python layers."""
        out = list(self.input_shapes.keys())
        for l in self.layers:
            if l.type in DATA_TYPES:
                out.extend(t for t in l.tops if t not in out)
        return out

    def infer(self, data_shapes: Optional[Dict[str, Shape]] = None) -> Dict[str, Shape]:
        """Compute every blob's NCHW shape; ``data_shapes`` supplies data-layer tops."""
        shapes: Dict[str, Shape] = dict(self.input_shapes)
        if data_shapes:
            shapes.update({k: tuple(int(d) for d in v) for k, v in data_shapes.items()})
        self.param_shapes = {}
        for l in self.layers:
            t = l.type
            if t in DATA_TYPES:
                for tp in l.tops:
                    if tp not in shapes:
                        raise KeyError("shape of data blob %r (layer %s) was not provided" % (tp, l.name))
                continue
            try:
                bots = [shapes[b] for b in l.bottoms]
            except KeyError as e:
                raise KeyError("layer %s: unknown bottom blob %s" % (l.name, e)) from None
            if t == "Convolution":
                p = l.sub("convolution_param")
                k, s, pad = kernel_stride_pad(p)
                g = int(p.get("group", 1))
                co = int(p.get("num_output"))
                n, c, h, w = bots[0]
                self.param_shapes[l.name] = [(co, c // g, k, k)] + ([(co,)] if bool(p.get("bias_term", True)) else [])
                shapes[l.tops[0]] = (n, co, conv_out(h, k, s, pad), conv_out(w, k, s, pad))
            elif t == "Deconvolution":
                p = l.sub("convolution_param")
                k, s, pad = kernel_stride_pad(p)
                g = int(p.get("group", 1))
                co = int(p.get("num_output"))
                n, c, h, w = bots[0]
                self.param_shapes[l.name] = [(c, co // g, k, k)] + ([(co,)] if bool(p.get("bias_term", True)) else [])
                shapes[l.tops[0]] = (n, co, deconv_out(h, k, s, pad), deconv_out(w, k, s, pad))
            elif t == "Pooling":
                p = l.sub("pooling_param")
                n, c, h, w = bots[0]
                if bool(p.get("global_pooling", False)):
                    shapes[l.tops[0]] = (n, c, 1, 1)
                else:
                    k, s, pad = kernel_stride_pad(p)
                    shapes[l.tops[0]] = (n, c, pool_out(h, k, s, pad), pool_out(w, k, s, pad))
            elif t == "Concat":
                axis = int(l.sub("concat_param").get("axis", l.sub("concat_param").get("concat_dim", 1)))
                if axis != 1:
                    raise NotImplementedError("Concat along axis %d" % axis)
                n, _, h, w = bots[0]
                for b in bots[1:]:
                    if (b[0], b[2], b[3]) != (n, h, w):
                        raise ValueError("layer %s: concat inputs disagree: %s" % (l.name, bots))
                shapes[l.tops[0]] = (n, sum(b[1] for b in bots), h, w)
            elif t == "Slice":
                sp = l.sub("slice_param")
                axis = int(sp.get("axis", sp.get("slice_dim", 1)))
                if axis != 1:
                    raise NotImplementedError("Slice along axis %d" % axis)
                n, c, h, w = bots[0]
                pts = [int(x) for x in sp.getall("slice_point")]
                if not pts:
                    step = c // len(l.tops)
                    pts = [step * i for i in range(1, len(l.tops))]
                edges = [0] + pts + [c]
                if len(edges) != len(l.tops) + 1 or any(b <= a for a, b in zip(edges[:-1], edges[1:])):
                    raise ValueError("layer %s: bad slice points %s for %d channels" % (l.name, pts, c))
                for tp, a, b in zip(l.tops, edges[:-1], edges[1:]):
                    shapes[tp] = (n, b - a, h, w)
            elif t in LOSS_TYPES:
                shapes[l.tops[0]] = ()
            elif t in ("ReLU", "Sigmoid", "Power", "LRN", "Dropout", "Softmax", "TanH"):
                shapes[l.tops[0]] = bots[0]
            elif t == "Eltwise":
                for b in bots[1:]:
                    if b != bots[0]:
                        raise ValueError("layer %s: eltwise inputs disagree: %s" % (l.name, bots))
                shapes[l.tops[0]] = bots[0]
            else:
                raise NotImplementedError("layer type %r (layer %s)" % (t, l.name))
        self.blob_shapes = shapes
        return shapes

    # ------------------------------------------------------------------
    def output_blobs(self) -> List[str]:
        """Tops that no later layer consumes (what ``net.forward()`` returns), in file order."""
        consumed = set()
        for l in self.layers:
            consumed.update(l.bottoms)
        out: List[str] = []
        for l in self.layers:
            for t in l.tops:
                if t not in consumed and t not in out:
                    out.append(t)
        return out

    def param_layers(self) -> List[Layer]:
        return [l for l in self.layers if l.name in self.param_shapes]


# ----------------------------------------------------------------------
# fillers (Caffe filler.hpp semantics)
# ----------------------------------------------------------------------

def bilinear_kernel(k: int) -> np.ndarray:
    f = int(math.ceil(k / 2.0))
    c = (2 * f - 1 - f % 2) / (2.0 * f)
    v = 1.0 - np.abs(np.arange(k) / float(f) - c)
    return np.outer(v, v).astype(np.float32)


def fill_blob(shape: Shape, filler: Optional[proto.Msg], rng: np.random.Generator) -> np.ndarray:
    ftype = str(filler.get("type", "constant")) if filler is not None else "constant"
    if ftype == "constant":
        return np.full(shape, float(filler.get("value", 0.0)) if filler is not None else 0.0, np.float32)
    if ftype == "xavier":
        fan_in = int(np.prod(shape)) // shape[0]
        fan_out = int(np.prod(shape)) // shape[1] if len(shape) > 1 else fan_in
        norm = str(filler.get("variance_norm", "FAN_IN"))
        n = fan_in if norm == "FAN_IN" else fan_out if norm == "FAN_OUT" else (fan_in + fan_out) / 2.0
        scale = math.sqrt(3.0 / n)
        return rng.uniform(-scale, scale, size=shape).astype(np.float32)
    if ftype == "gaussian":
        return (rng.standard_normal(size=shape) * float(filler.get("std", 1.0)) + float(filler.get("mean", 0.0))).astype(np.float32)
    if ftype == "uniform":
        return rng.uniform(float(filler.get("min", 0.0)), float(filler.get("max", 1.0)), size=shape).astype(np.float32)
    if ftype == "bilinear":
        if len(shape) != 4 or shape[2] != shape[3]:
            raise ValueError("bilinear filler needs a square 4-d blob")
        return np.broadcast_to(bilinear_kernel(shape[3]), shape).astype(np.float32).copy()
    raise NotImplementedError("filler type %r" % ftype)


def fill_params(spec: NetSpec, seed: int = 0) -> Dict[str, List[np.ndarray]]:
    """Seeded filler initialisation of every learnable blob, in layer order."""
    rng = np.random.default_rng(seed)
    out: Dict[str, List[np.ndarray]] = {}
    for l in spec.param_layers():
        p = l.sub("convolution_param")
        shapes = spec.param_shapes[l.name]
        blobs = [fill_blob(shapes[0], p.get("weight_filler"), rng)]
        if len(shapes) > 1:
            blobs.append(fill_blob(shapes[1], p.get("bias_filler"), rng))
        out[l.name] = blobs
    return out
