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
                    if self.phase == "TRAIN":      # the argmax is what backward routes the gradient through
                        y, idx = R.max_pool(bots[0], k, s, pad, return_index=True)
                        self.aux[name] = idx
                    else:
                        y = R.max_pool(bots[0], k, s, pad)
                    B[tops[0]] = y
                else:
                    B[tops[0]] = R.ave_pool(bots[0], k, s, pad)
            elif t == "LRN":
                p = l.get("lrn_param")
                args = (bots[0], int(p.get("local_size", 5)), float(p.get("alpha", 1.0)), float(p.get("beta", 0.75)), float(p.get("k", 1.0)))
                if self.phase == "TRAIN":      # the scale is what backward needs
                    y, scale = R.lrn_across(*args, return_scale=True)
                    self.aux[name] = scale
                else:
                    y = R.lrn_across(*args)
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
                    # counter-based mask shared bit-for-bit with the HIP kernel (R.dropout_mask)
                    mask = R.dropout_mask(bots[0].shape, ratio, getattr(self, "dropout_seed", 0))
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
            hook = getattr(self, "round_activations", None)      # models reduced-precision storage of the activations
            if hook is not None:
                for tp in tops:
                    if tp in B and getattr(B[tp], "ndim", 0) == 4:
                        B[tp] = hook(tp, B[tp])
        return B

    # -- backward -----------------------------------------------------------
    def loss_weights(self) -> Dict[str, float]:
        out = {}
        for l in self.layers:
            t = l.get("type")
            lw = [float(v) for v in l.getall("loss_weight")]
            for i, tp in enumerate(l.getall("top")):
                w = lw[i] if i < len(lw) else (1.0 if (t in ("L1Loss", "EuclideanLoss", "SoftmaxWithLoss") and i == 0) else 0.0)
                if w:
                    out[tp] = w
        return out

    def total_loss(self) -> float:
        return float(sum(w * self.losses[k] for k, w in self.loss_weights().items()))

    def backward(self, stop_at: Optional[str] = None) -> Dict[str, List[np.ndarray]]:
        """Net::Backward: returns {layer: [dW, db]}; blob gradients are left in self.diffs.  `stop_at`: the last layer
        (walking backwards) whose gradients are wanted - full-size checks only need the top of the net."""
        B = self.blobs
        D: Dict[str, np.ndarray] = {}
        grads: Dict[str, List[np.ndarray]] = {}
        self._bwd_stop = False
        lw = self.loss_weights()
        data_tops = set(self.inputs)
        for l in self.layers:
            if l.get("type") in ("Data", "Python", "Input"):
                data_tops.update(l.getall("top"))

        def acc(name, g):
            if name in data_tops:
                return
            if name in D:
                D[name] = D[name] + g
            else:
                D[name] = g.astype(F32)

        for l in reversed(self.layers):
            t = l.get("type")
            name = l.get("name")
            bots = l.getall("bottom")
            tops = l.getall("top")
            if t in ("Data", "Python", "Input"):
                continue
            if t in ("L1Loss", "EuclideanLoss"):
                w = lw.get(tops[0], 0.0)
                a, b = B[bots[0]], B[bots[1]]
                g = R.l1_loss_grad(a, b, w) if t == "L1Loss" else R.euclidean_loss_grad(a, b, w)
                acc(bots[0], g)
                acc(bots[1], -g)
                continue
            if t == "SoftmaxWithLoss":
                lp = l.get("loss_param")
                norm = bool(lp.get("normalize", True)) if lp else True
                ign = lp.get("ignore_label", None) if lp else None
                acc(bots[0], R.softmax_loss_grad(B[bots[0]], B[bots[1]], norm, ign, lw.get(tops[0], 0.0)))
                continue
            if stop_at is not None and getattr(self, "_bwd_stop", False):
                break
            if name == stop_at:
                self._bwd_stop = True      # this layer is still processed, the ones below it are not
            if tops[0] not in D:
                continue   # nothing flows back through this layer
            dy = D[tops[0]]
            if t == "Convolution":
                p = l.get("convolution_param")
                k, s, pad = _ksp(p)
                need_dx = bots[0] not in data_tops and self._needs_grad(bots[0], data_tops)
                dw, db, dx = R.conv2d_backward(B[bots[0]], self.params[name][0], dy, pad, s, need_dx)
                grads[name] = [dw] + ([db] if len(self.params[name]) > 1 else [])
                if need_dx:
                    acc(bots[0], dx)
            elif t == "Deconvolution":
                # the reference freezes its (bilinear) deconvolutions (lr_mult 0): only the data gradient is restated
                p = l.get("convolution_param")
                k, s, pad = _ksp(p)
                grads[name] = [np.zeros_like(a) for a in self.params[name]]
                if bots[0] not in data_tops and self._needs_grad(bots[0], data_tops):
                    acc(bots[0], R.deconv2d_backward_data(dy, self.params[name][0], pad, s, int(p.get("group", 1))))
            elif t == "ReLU":
                g = dy * (B[tops[0]] > 0)
                if bots[0] == tops[0]:
                    D[bots[0]] = g.astype(F32)
                else:
                    acc(bots[0], g)
            elif t == "Sigmoid":
                acc(bots[0], R.sigmoid_backward(B[tops[0]], dy))
            elif t == "Power":
                p = l.get("power_param")
                if float(p.get("power", 1.0)) != 1.0:
                    raise NotImplementedError("oracle backward: Power with power != 1")
                acc(bots[0], dy * F32(float(p.get("scale", 1.0))))
            elif t == "Pooling":
                p = l.get("pooling_param")
                if p.get("pool", "MAX") != "MAX":
                    raise NotImplementedError("oracle backward: AVE pooling")
                acc(bots[0], R.max_pool_backward(dy, self.aux[name], B[bots[0]].shape))
            elif t == "LRN":
                p = l.get("lrn_param")
                acc(bots[0], R.lrn_across_backward(B[bots[0]], B[tops[0]], self.aux[name], dy, int(p.get("local_size", 5)),
                                                   float(p.get("alpha", 1.0)), float(p.get("beta", 0.75))))
            elif t == "Concat":
                off = 0
                for b in bots:
                    c = B[b].shape[1]
                    acc(b, dy[:, off:off + c])
                    off += c
            elif t == "Slice":
                pass   # handled below (needs all tops)
            elif t == "Dropout":
                ratio = float(l.get("dropout_param").get("dropout_ratio", 0.5))
                if self.phase == "TEST":
                    acc(bots[0], dy)
                else:
                    acc(bots[0], dy * self.dropout_masks[name] * F32(1.0 / (1.0 - ratio)))
            elif t == "Eltwise":
                p = l.get("eltwise_param")
                op = p.get("operation", "SUM") if p else "SUM"
                if op == "PROD":
                    for i, b in enumerate(bots):
                        other = None
                        for j, b2 in enumerate(bots):
                            if j != i:
                                other = B[b2] if other is None else other * B[b2]
                        acc(b, dy * other)
                elif op == "SUM":
                    cf = [float(c) for c in p.getall("coeff")] if p and p.getall("coeff") else [1.0] * len(bots)
                    for c_, b in zip(cf, bots):
                        acc(b, dy * F32(c_))
                else:
                    raise NotImplementedError("oracle backward: Eltwise MAX")
            else:
                raise NotImplementedError("oracle backward: %s" % t)
        self.diffs = D
        return grads

    def _needs_grad(self, blob: str, data_tops) -> bool:
        """True if some learnable layer lies upstream of `blob` (Caffe's propagate_down)."""
        produced = {}
        for l in self.layers:
            for tp in l.getall("top"):
                produced.setdefault(tp, []).append(l)
        seen, stack = set(), [blob]
        while stack:
            b = stack.pop()
            if b in seen or b in data_tops:
                continue
            seen.add(b)
            for l in produced.get(b, []):
                if l.get("name") in self.params:
                    return True
                stack.extend(l.getall("bottom"))
        return False


class RefSolver:
    """Caffe Solver::Step restated: ForwardBackward, Regularize + ComputeUpdateValue (SGD or Adam), Update."""

    def __init__(self, net: "RefNet", solver_msg, lr_mults: Dict[str, List[float]], decay_mults: Dict[str, List[float]]):
        self.net = net
        g = solver_msg.get
        self.base_lr = float(g("base_lr", 0.01))
        self.momentum = float(g("momentum", 0.0))
        self.momentum2 = float(g("momentum2", 0.999))
        self.delta = float(g("delta", 1e-8))
        self.weight_decay = float(g("weight_decay", 0.0))
        self.lr_policy = str(g("lr_policy", "fixed"))
        self.gamma = float(g("gamma", 0.1))
        self.stepsize = int(g("stepsize", 1))
        st = g("solver_type", g("type", "SGD"))
        self.kind = str(st).upper()
        self.iter = 0
        self.lr_mults, self.decay_mults = lr_mults, decay_mults
        self.hist = {k: [np.zeros_like(b) for b in v] for k, v in net.params.items()}
        self.hist2 = {k: [np.zeros_like(b) for b in v] for k, v in net.params.items()}

    def lr(self) -> float:
        if self.lr_policy == "fixed":
            return self.base_lr
        if self.lr_policy == "step":
            return self.base_lr * self.gamma ** (self.iter // self.stepsize)
        raise NotImplementedError(self.lr_policy)

    def apply(self, grads: Dict[str, List[np.ndarray]]) -> None:
        rate = self.lr()
        for name, gs in grads.items():
            for i, g in enumerate(gs):
                lm = self.lr_mults.get(name, [1.0, 1.0])
                dm = self.decay_mults.get(name, [1.0, 1.0])
                lmi = lm[i] if i < len(lm) else 1.0
                dmi = dm[i] if i < len(dm) else 1.0
                w = self.net.params[name][i]
                if lmi == 0.0:
                    continue
                if self.kind == "ADAM":
                    R.adam_update(w, g, self.hist[name][i], self.hist2[name][i], rate, self.momentum, self.momentum2, self.delta,
                                  self.weight_decay, lmi, dmi, self.iter + 1)
                else:
                    R.sgd_update(w, g, self.hist[name][i], rate, self.momentum, self.weight_decay, lmi, dmi)
        self.iter += 1
