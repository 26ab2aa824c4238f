"""ORACLE — TEST INFRASTRUCTURE ONLY.  Parity unpinned (see oracle/caffe_ref.py).

Layer-by-layer CPU executor for a Caffe NetParameter, i.e. the restatement of
``caffe.Net.forward()`` / ``Net::ForwardBackward`` as the reference calls them
(reference: scripts/fcn_object_detector.py:87, train/train.sh:25-28) on the
reference's own prototxt files.  It shares only the text tokenizer
(fcn_object_detector_amd.proto) with the shipped package; shape rules and
layer arithmetic are its own (oracle/caffe_ref.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np

from . import caffe_ref as R

F32 = np.float32


def _phase_ok(layer, phase: str) -> bool:
    inc = layer.getall("include")
    if inc:
        return any(m.get("phase", phase) == phase for m in inc)
    exc = layer.getall("exclude")
    if exc:
        return not any(m.get("phase", None) == phase for m in exc)
    return True


def _ksp(p, default_stride=1):
    k = p.get("kernel_size", None)
    if k is None:
        k = p.get("kernel_h")
    return int(k), int(p.get("stride", default_stride)), int(p.get("pad", 0))


class RefNet:
    """Executes the layers of a parsed prototxt in file order on numpy NCHW float32 blobs."""

    def __init__(self, net_msg, phase: str = "TEST", params: Optional[Dict[str, List[np.ndarray]]] = None,
                 input_shapes: Optional[Dict[str, tuple]] = None):
        self.phase = phase
        self.layers = [l for l in net_msg.getall("layer") if _phase_ok(l, phase)]
        self.params: Dict[str, List[np.ndarray]] = params if params is not None else {}
        self.blobs: Dict[str, np.ndarray] = {}
        self.inputs: List[str] = []
        names = net_msg.getall("input")
        shapes = net_msg.getall("input_shape")
        dims = net_msg.getall("input_dim")
        for i, nm in enumerate(names):
            if input_shapes and nm in input_shapes:
                shp = tuple(input_shapes[nm])
            elif shapes:
                shp = tuple(int(d) for d in shapes[i].getall("dim"))
            else:
                shp = tuple(int(d) for d in dims[4 * i:4 * i + 4])
            self.blobs[nm] = np.zeros(shp, F32)
            self.inputs.append(nm)
        self.dropout_masks: Dict[str, np.ndarray] = {}
        self.aux: Dict[str, object] = {}
        self.losses: Dict[str, float] = {}

    # -- parameter shapes (what a filler must produce) -------------------
    def param_shapes(self, input_shapes: Dict[str, tuple]) -> Dict[str, List[tuple]]:
        """Dry-run shapes: {layer: [W shape, b shape]} given the input blob shapes."""
        shapes = dict(input_shapes)
        out: Dict[str, List[tuple]] = {}
        for l in self.layers:
            t = l.get("type")
            bots = [shapes[b] for b in l.getall("bottom")]
            tops = l.getall("top")
            if t == "Convolution":
                p = l.get("convolution_param")
                k, s, pad = _ksp(p)
                g = int(p.get("group", 1))
                co = int(p.get("num_output"))
                n, c, h, w = bots[0]
                out[l.get("name")] = [(co, c // g, k, k)] + ([(co,)] if p.get("bias_term", True) else [])
                shapes[tops[0]] = (n, co, R.conv_out(h, k, pad, s), R.conv_out(w, k, pad, s))
            elif t == "Deconvolution":
                p = l.get("convolution_param")
                k, s, pad = _ksp(p)
                g = int(p.get("group", 1))
                co = int(p.get("num_output"))
                n, c, h, w = bots[0]
                out[l.get("name")] = [(c, co // g, k, k)] + ([(co,)] if p.get("bias_term", True) else [])
                shapes[tops[0]] = (n, co, R.deconv_out(h, k, pad, s), R.deconv_out(w, k, pad, s))
            elif t == "Pooling":
                p = l.get("pooling_param")
                k, s, pad = _ksp(p)
                n, c, h, w = bots[0]
                shapes[tops[0]] = (n, c, R.pool_out(h, k, pad, s), R.pool_out(w, k, pad, s))
            elif t == "Concat":
                n, _, h, w = bots[0]
                shapes[tops[0]] = (n, sum(b[1] for b in bots), h, w)
            elif t == "Slice":
                pts = [int(x) for x in l.get("slice_param").getall("slice_point")]
                n, c, h, w = bots[0]
                edges = [0] + pts + [c]
                for tp, a, b in zip(tops, edges[:-1], edges[1:]):
                    shapes[tp] = (n, b - a, h, w)
            elif t in ("L1Loss", "EuclideanLoss", "SoftmaxWithLoss"):
                shapes[tops[0]] = ()
            elif t in ("Data", "Python", "Input"):
                for tp in tops:
                    if tp not in shapes:
                        raise KeyError("shape of data top %r must be supplied" % tp)
            else:
                for tp in tops:
                    shapes[tp] = bots[0]
        self.shapes = shapes
        return out

    # -- forward ----------------------------------------------------------
    def forward(self, dropout_rng: Optional[np.random.Generator] = None) -> Dict[str, np.ndarray]:
        B = self.blobs
        for l in self.layers:
            t = l.get("type")
            name = l.get("name")
            bots = [B[b] for b in l.getall("bottom")]
            tops = l.getall("top")
            if t in ("Data", "Python", "Input"):
                continue  # tops are provided by the caller through self.blobs
            if t == "Convolution":
                p = l.get("convolution_param")
                k, s, pad = _ksp(p)
                w = self.params[name][0]
                b = self.params[name][1] if len(self.params[name]) > 1 else None
                B[tops[0]] = R.conv2d(bots[0], w, b, pad, s, int(p.get("group", 1)))
            elif t == "Deconvolution":
                p = l.get("convolution_param")
                k, s, pad = _ksp(p)
                w = self.params[name][0]
                b = self.params[name][1] if len(self.params[name]) > 1 else None
                B[tops[0]] = R.deconv2d(bots[0], w, b, pad, s, int(p.get("group", 1)))
            elif t == "ReLU":
                rp = l.get("relu_param")
                B[tops[0]] = R.relu(bots[0], float(rp.get("negative_slope", 0.0)) if rp else 0.0)
            elif t == "Sigmoid":
                B[tops[0]] = R.sigmoid(bots[0])
            elif t == "Power":
                p = l.get("power_param")
                B[tops[0]] = R.power(bots[0], float(p.get("power", 1.0)), float(p.get("scale", 1.0)),
                                     float(p.get("shift", 0.0)))
            elif t == "Pooling":
                p = l.get("pooling_param")
                k, s, pad = _ksp(p)
                if p.get("pool", "MAX") == "MAX":
                    y, idx = R.max_pool(bots[0], k, s, pad, return_index=True)
                    self.aux[name] = idx
                    B[tops[0]] = y
                else:
                    B[tops[0]] = R.ave_pool(bots[0], k, s, pad)
            elif t == "LRN":
                p = l.get("lrn_param")
                y, scale = R.lrn_across(bots[0], int(p.get("local_size", 5)), float(p.get("alpha", 1.0)),
                                        float(p.get("beta", 0.75)), float(p.get("k", 1.0)), return_scale=True)
                self.aux[name] = scale
                B[tops[0]] = y
            elif t == "Concat":
                B[tops[0]] = np.concatenate(bots, axis=1)
            elif t == "Slice":
                pts = [int(x) for x in l.get("slice_param").getall("slice_point")]
                edges = [0] + pts + [bots[0].shape[1]]
                for tp, a, b in zip(tops, edges[:-1], edges[1:]):
                    B[tp] = np.ascontiguousarray(bots[0][:, a:b])
            elif t == "Dropout":
                ratio = float(l.get("dropout_param").get("dropout_ratio", 0.5))
                if self.phase == "TEST":
                    B[tops[0]] = bots[0]
                else:
                    mask = self.dropout_masks.get(name)
                    if mask is None:
                        rng = dropout_rng or np.random.default_rng(0)
                        mask = (rng.random(bots[0].shape) >= ratio).astype(F32)
                        self.dropout_masks[name] = mask
                    B[tops[0]] = bots[0] * mask * F32(1.0 / (1.0 - ratio))
            elif t == "Eltwise":
                p = l.get("eltwise_param")
                op = p.get("operation", "SUM") if p else "SUM"
                B[tops[0]] = R.eltwise(bots, op, p.getall("coeff") if p else None)
            elif t == "Softmax":
                B[tops[0]] = R.softmax(bots[0], 1)
            elif t == "L1Loss":
                self.losses[tops[0]] = R.l1_loss(bots[0], bots[1])
                B[tops[0]] = np.array(self.losses[tops[0]], F32)
            elif t == "EuclideanLoss":
                self.losses[tops[0]] = R.euclidean_loss(bots[0], bots[1])
                B[tops[0]] = np.array(self.losses[tops[0]], F32)
            elif t == "SoftmaxWithLoss":
                lp = l.get("loss_param")
                norm = bool(lp.get("normalize", True)) if lp else True
                ign = lp.get("ignore_label", None) if lp else None
                self.losses[tops[0]] = R.softmax_loss(bots[0], bots[1], norm, ign)
                B[tops[0]] = np.array(self.losses[tops[0]], F32)
            else:
                raise NotImplementedError("oracle: layer type %r (%s)" % (t, name))
        return B
