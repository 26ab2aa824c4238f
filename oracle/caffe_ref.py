"""ORACLE — TEST INFRASTRUCTURE ONLY.  Parity unpinned (see below).

CPU restatement (numpy / OpenBLAS) of the Caffe layer semantics that the
reference's hot path executes inside an external, un-vendored NVIDIA-Caffe
install (reference: scripts/fcn_object_detector.py:87 ``net.forward()`` on
models/deploy.prototxt; train/train.sh:25-28 ``caffe train``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the shipped package ``fcn_object_detector_amd`` never does.

PARITY UNPINNED: the reference has no tests, golden vectors or fixtures
 and neither Caffe nor OpenCV can be run here, so this file
restates the *published* BVLC/NVIDIA-Caffe algorithms layer by layer — the same
schedule Caffe's CPU path runs (im2col + sgemm per conv, separate bias add,
ReLU pass, ceil-mode pooling, across-channel LRN, concat copy).  Its layer ops
are cross-checked against torch CPU functional ops in tests/test_oracle.py.

All tensors are NCHW float32, like pycaffe blobs.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import ctypes
import os

import numpy as np

F32 = np.float32


def _load_c():
    """oracle/libcaffe_cpu.so (oracle/Makefile, built by __graft_entry__.build()): compiled loops for im2col, MAX pooling
    and LRN.  Absent or ORACLE_NUMPY=1: the numpy statements below run instead (tests/test_oracle.py holds the two equal)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcaffe_cpu.so")
    if os.environ.get("ORACLE_NUMPY") or not os.path.isfile(path):
        return None
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")      # read when libgomp loads: idle OpenMP threads sleep instead of spinning
    lib = ctypes.CDLL(path)
    i, f, p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p
    lib.oracle_im2col_f32.argtypes = [p] + [i] * 11 + [p]
    lib.oracle_maxpool_f32.argtypes = [p] + [i] * 8 + [p, p]
    lib.oracle_lrn_f32.argtypes = [p] + [i] * 5 + [f, f, f, p, p]
    for fn in (lib.oracle_im2col_f32, lib.oracle_maxpool_f32, lib.oracle_lrn_f32):
        fn.restype = None
    return lib


_C = _load_c()


def _ptr(a: np.ndarray) -> int:
    assert a.flags.c_contiguous
    return a.ctypes.data


# --------------------------------------------------------------------------
# shape rules
# --------------------------------------------------------------------------

def conv_out(h: int, k: int, p: int, s: int) -> int:
    """Caffe ConvolutionLayer::compute_output_shape: floor((H + 2p - k) / s) + 1."""
    return (h + 2 * p - k) // s + 1


def pool_out(h: int, k: int, p: int, s: int) -> int:
    """Caffe PoolingLayer::Reshape: ceil((H + 2p - k) / s) + 1, minus one if the
    last window would start in the bottom/right padding."""
    o = int(math.ceil((h + 2 * p - k) / float(s))) + 1
    if p > 0 and (o - 1) * s >= h + p:
        o -= 1
    return o


def deconv_out(h: int, k: int, p: int, s: int) -> int:
    """Caffe DeconvolutionLayer::compute_output_shape: s (H - 1) + k - 2p."""
    return s * (h - 1) + k - 2 * p


# --------------------------------------------------------------------------
# forward ops
# --------------------------------------------------------------------------

def im2col(x: np.ndarray, kh: int, kw: int, ph: int, pw: int, sh: int, sw: int) -> np.ndarray:
    """(C,H,W) -> (C*kh*kw, OH*OW), zero padding, row order (c, r, q) as Caffe's im2col_cpu."""
    c, h, w = x.shape
    oh, ow = conv_out(h, kh, ph, sh), conv_out(w, kw, pw, sw)
    if _C is not None and x.dtype == F32:
        x = np.ascontiguousarray(x)
        col = np.empty((c * kh * kw, oh * ow), dtype=F32)
        _C.oracle_im2col_f32(_ptr(x), c, h, w, kh, kw, ph, pw, sh, sw, oh, ow, _ptr(col))
        return col
    xp = np.zeros((c, h + 2 * ph, w + 2 * pw), dtype=x.dtype)
    xp[:, ph:ph + h, pw:pw + w] = x
    s0, s1, s2 = xp.strides
    win = np.lib.stride_tricks.as_strided(
        xp, shape=(c, kh, kw, oh, ow), strides=(s0, s1, s2, s1 * sh, s2 * sw), writeable=False)
    return np.ascontiguousarray(win).reshape(c * kh * kw, oh * ow)


def conv2d(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray], pad: int, stride: int, group: int = 1) -> np.ndarray:
    """Caffe Convolution forward: per image, per group: W_g (Cout/g, Cin/g*kh*kw) @ im2col; then + bias."""
    n, cin, h, wd = x.shape
    cout, cin_g, kh, kw = w.shape
    assert cin_g * group == cin and cout % group == 0
    oh, ow = conv_out(h, kh, pad, stride), conv_out(wd, kw, pad, stride)
    y = np.empty((n, cout, oh, ow), dtype=F32)
    cog = cout // group
    for i in range(n):
        for g in range(group):
            if kh == 1 and kw == 1 and pad == 0 and stride == 1:      # Caffe's is_1x1_: the input IS the column buffer
                col = x[i, g * cin_g:(g + 1) * cin_g].reshape(cin_g, h * wd)
            else:
                col = im2col(x[i, g * cin_g:(g + 1) * cin_g], kh, kw, pad, pad, stride, stride)
            wg = w[g * cog:(g + 1) * cog].reshape(cog, cin_g * kh * kw)
            y[i, g * cog:(g + 1) * cog] = (wg @ col).reshape(cog, oh, ow)
    if b is not None:
        y += b.reshape(1, cout, 1, 1)
    return y


def deconv2d(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray], pad: int, stride: int, group: int = 1) -> np.ndarray:
    """Caffe Deconvolution forward: col = W_g^T @ x_g, then col2im (scatter-add).
    Weight blob shape (Cin, Cout/g, kh, kw)."""
    n, cin, h, wd = x.shape
    cin_w, cog, kh, kw = w.shape
    assert cin_w == cin and cin % group == 0
    cout = cog * group
    cig = cin // group
    oh, ow = deconv_out(h, kh, pad, stride), deconv_out(wd, kw, pad, stride)
    y = np.zeros((n, cout, oh + 2 * pad, ow + 2 * pad), dtype=F32)
    for i in range(n):
        for g in range(group):
            wg = w[g * cig:(g + 1) * cig].reshape(cig, cog * kh * kw)
            col = (wg.T @ x[i, g * cig:(g + 1) * cig].reshape(cig, h * wd)).reshape(cog, kh, kw, h, wd)
            for r in range(kh):
                for q in range(kw):
                    y[i, g * cog:(g + 1) * cog, r:r + stride * h:stride, q:q + stride * wd:stride] += col[:, r, q]
    y = y[:, :, pad:pad + oh, pad:pad + ow]
    if b is not None:
        y = y + b.reshape(1, cout, 1, 1)
    return np.ascontiguousarray(y, dtype=F32)


def relu(x: np.ndarray, negative_slope: float = 0.0) -> np.ndarray:
    return np.maximum(x, 0) + F32(negative_slope) * np.minimum(x, 0) if negative_slope else np.maximum(x, 0)


def sigmoid(x: np.ndarray) -> np.ndarray:
    """Caffe SigmoidLayer: 0.5 * tanh(0.5 x) + 0.5 (== 1 / (1 + exp(-x)))."""
    return (0.5 * np.tanh(0.5 * x.astype(np.float64)) + 0.5).astype(F32)


def power(x: np.ndarray, power_: float = 1.0, scale: float = 1.0, shift: float = 0.0) -> np.ndarray:
    """Caffe PowerLayer: (shift + scale * x) ^ power."""
    y = x * F32(scale) + F32(shift)
    return y if power_ == 1.0 else np.power(y, F32(power_))


def max_pool(x: np.ndarray, k: int, s: int, p: int, return_index: bool = False):
    """Caffe PoolingLayer MAX: window clipped to the image, first maximum in raster order wins
    (`>` compare, initial -FLT_MAX).  One pass per window element (r, q) in raster order over strided VIEWS of the
    input and output (no gathers: this runs inside the CPU baseline bench.py times)."""
    n, c, h, w = x.shape
    oh, ow = pool_out(h, k, p, s), pool_out(w, k, p, s)
    if _C is not None and x.dtype == F32:
        x = np.ascontiguousarray(x)
        y = np.empty((n, c, oh, ow), dtype=F32)
        idx = np.empty((n, c, oh, ow), dtype=np.int64) if return_index else None
        _C.oracle_maxpool_f32(_ptr(x), n * c, h, w, k, s, p, oh, ow, _ptr(y), _ptr(idx) if return_index else None)
        return (y, idx) if return_index else y
    # channels innermost while pooling: the window strides then sit on outer dimensions and every compare / masked copy
    # runs over contiguous channel runs
    xt = np.ascontiguousarray(x.transpose(0, 2, 3, 1))
    y = np.full((n, oh, ow, c), -np.finfo(F32).max, dtype=F32)
    idx = np.full((n, oh, ow, c), -1, dtype=np.int64) if return_index else None

    def span(r, size, out):
        # output positions o in [o0, o1) whose window element r falls inside the image: 0 <= o * s - p + r < size
        o0 = max(0, -((r - p) // s))
        o1 = min(out, (size - 1 + p - r) // s + 1)
        return o0, o1

    for r in range(k):
        oy0, oy1 = span(r, h, oh)
        if oy1 <= oy0:
            continue
        iy0 = oy0 * s - p + r
        for q in range(k):
            ox0, ox1 = span(q, w, ow)
            if ox1 <= ox0:
                continue
            ix0 = ox0 * s - p + q
            cand = xt[:, iy0:iy0 + (oy1 - oy0 - 1) * s + 1:s, ix0:ix0 + (ox1 - ox0 - 1) * s + 1:s, :]
            cur = y[:, oy0:oy1, ox0:ox1, :]
            if not return_index:
                # values only (TEST phase, the CPU baseline): the running maximum, vectorised.  Same numbers as the strict
                # `>` update below for finite inputs (which of two equal zeros survives is the only freedom, and no layer of
                # these nets can tell them apart)
                np.maximum(cur, cand, out=cur)
                continue
            take = cand > cur
            np.copyto(cur, cand, where=take)
            if return_index:
                flat = (np.arange(iy0, iy0 + (oy1 - oy0 - 1) * s + 1, s)[:, None] * w +
                        np.arange(ix0, ix0 + (ox1 - ox0 - 1) * s + 1, s)[None, :])[None, :, :, None]
                np.copyto(idx[:, oy0:oy1, ox0:ox1, :], np.broadcast_to(flat, take.shape), where=take)
    y = np.ascontiguousarray(y.transpose(0, 3, 1, 2))
    return (y, np.ascontiguousarray(idx.transpose(0, 3, 1, 2))) if return_index else y


def ave_pool(x: np.ndarray, k: int, s: int, p: int) -> np.ndarray:
    """Caffe PoolingLayer AVE: divisor is the window area clipped to H+p (includes padding),
    the sum runs over the part inside the image."""
    n, c, h, w = x.shape
    oh, ow = pool_out(h, k, p, s), pool_out(w, k, p, s)
    y = np.zeros((n, c, oh, ow), dtype=F32)
    for py in range(oh):
        hs = py * s - p
        he = min(hs + k, h + p)
        for px in range(ow):
            ws = px * s - p
            we = min(ws + k, w + p)
            size = (he - hs) * (we - ws)
            a, b_, c_, d = max(hs, 0), min(he, h), max(ws, 0), min(we, w)
            y[:, :, py, px] = x[:, :, a:b_, c_:d].sum(axis=(2, 3), dtype=F32) / F32(size)
    return y


def lrn_across(x: np.ndarray, local_size: int, alpha: float, beta: float, k: float = 1.0,
               return_scale: bool = False):
    """Caffe LRNLayer ACROSS_CHANNELS: scale = k + alpha/n * sum_{c' in window} x^2 (zero padded),
    y = x * scale^-beta."""
    n, c, h, w = x.shape
    if _C is not None and x.dtype == F32:
        x = np.ascontiguousarray(x)
        y, scale = np.empty_like(x), np.empty_like(x)
        _C.oracle_lrn_f32(_ptr(x), n, c, h, w, local_size, F32(alpha / local_size), beta, k, _ptr(y), _ptr(scale))
        if return_scale:
            # TRAIN phase (backward reads the scale; the committed training fixtures were made by the numpy statement): the
            # compiled loop's scale is bit-identical to numpy's, its powf is not (a last-bit difference) - so the power is numpy's
            return (x * np.power(scale, F32(-beta))).astype(F32), scale
        return y
    pre = (local_size - 1) // 2
    sq = np.zeros((n, c + local_size - 1, h, w), dtype=F32)
    sq[:, pre:pre + c] = x * x
    acc = np.zeros_like(x)
    for j in range(local_size):
        acc += sq[:, j:j + c]
    scale = F32(k) + F32(alpha / local_size) * acc
    y = (x * np.power(scale, F32(-beta))).astype(F32)
    return (y, scale) if return_scale else y


def softmax(x: np.ndarray, axis: int = 1) -> np.ndarray:
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(axis=axis, keepdims=True)).astype(F32)


def eltwise(xs: Sequence[np.ndarray], op: str = "SUM", coeff: Optional[Sequence[float]] = None) -> np.ndarray:
    if op == "PROD":
        y = xs[0] * xs[1]
        for t in xs[2:]:
            y = y * t
        return y.astype(F32)
    if op == "SUM":
        cf = list(coeff) if coeff else [1.0] * len(xs)
        y = F32(cf[0]) * xs[0]
        for c_, t in zip(cf[1:], xs[1:]):
            y = y + F32(c_) * t
        return y.astype(F32)
    if op == "MAX":
        y = xs[0]
        for t in xs[1:]:
            y = np.maximum(y, t)
        return y.astype(F32)
    raise ValueError(op)


# --------------------------------------------------------------------------
# losses (forward value, and gradient w.r.t. the first bottom for loss_weight w)
# --------------------------------------------------------------------------

def l1_loss(a: np.ndarray, b: np.ndarray) -> float:
    """NVIDIA-Caffe L1LossLayer: sum |a - b| / num."""
    return float(np.abs(a.astype(np.float64) - b).sum() / a.shape[0])


def l1_loss_grad(a: np.ndarray, b: np.ndarray, w: float = 1.0) -> np.ndarray:
    return (np.sign(a - b) * F32(w / a.shape[0])).astype(F32)


def euclidean_loss(a: np.ndarray, b: np.ndarray) -> float:
    """Caffe EuclideanLossLayer: sum (a - b)^2 / (2 num)."""
    d = a.astype(np.float64) - b
    return float((d * d).sum() / (2.0 * a.shape[0]))


def euclidean_loss_grad(a: np.ndarray, b: np.ndarray, w: float = 1.0) -> np.ndarray:
    return ((a - b) * F32(w / a.shape[0])).astype(F32)


def _softmax_loss_terms(x: np.ndarray, label: np.ndarray, ignore_label: Optional[int]):
    p = softmax(x, 1)
    n, c = x.shape[:2]
    lab = label.reshape(n, 1, -1).astype(np.int64)
    valid = np.ones(lab.shape, bool) if ignore_label is None else (lab != ignore_label)
    lab_c = np.clip(lab, 0, c - 1)
    return p, lab_c, valid


def softmax_loss(x: np.ndarray, label: np.ndarray, normalize: bool = True,
                 ignore_label: Optional[int] = None) -> float:
    """Caffe SoftmaxWithLoss (legacy `normalize` flag): -sum log p[label] / (count_valid if normalize else N)
    (softmax_loss_layer.cpp Forward_cpu: prob clamped at FLT_MIN before the log)."""
    p, lab, valid = _softmax_loss_terms(x, label, ignore_label)
    n, c = x.shape[:2]
    pl = np.take_along_axis(p.reshape(n, c, -1), lab, axis=1)
    terms = -np.log(np.maximum(pl, np.finfo(F32).tiny).astype(F32)).astype(np.float64)
    cnt = int(valid.sum())
    return float((terms * valid).sum() / (max(cnt, 1) if normalize else n))


def softmax_loss_grad(x: np.ndarray, label: np.ndarray, normalize: bool = True, ignore_label: Optional[int] = None,
                      w: float = 1.0) -> np.ndarray:
    """Backward_cpu: (p - onehot(label)) * loss_weight / denom, zero at ignored pixels."""
    p, lab, valid = _softmax_loss_terms(x, label, ignore_label)
    n, c = x.shape[:2]
    g = p.reshape(n, c, -1).copy()
    np.put_along_axis(g, lab, np.take_along_axis(g, lab, axis=1) - F32(1.0), axis=1)
    g = g * valid
    denom = max(int(valid.sum()), 1) if normalize else n
    return (g * F32(w / denom)).reshape(x.shape).astype(F32)


def deconv2d_backward_data(dy: np.ndarray, w: np.ndarray, pad: int, stride: int, group: int = 1) -> np.ndarray:
    """Deconvolution backward w.r.t. the input = the forward convolution of dY with the same blob
    (deconv weights are (Cin, Cout/g, kh, kw) = a conv bank with num_output Cin)."""
    return conv2d(dy, w, None, pad, stride, group)


# --------------------------------------------------------------------------
# fillers (Caffe filler.hpp)
# --------------------------------------------------------------------------

def bilinear_filler(shape: Tuple[int, int, int, int]) -> np.ndarray:
    """Caffe BilinearFiller: f = ceil(k/2), c = (2f - 1 - f%2) / (2f), w[y,x] = (1-|x/f-c|)(1-|y/f-c|)."""
    n, c, kh, kw = shape
    assert kh == kw
    f = int(math.ceil(kw / 2.0))
    cc = (2 * f - 1 - f % 2) / (2.0 * f)
    xs = np.arange(kw)
    k1 = 1 - np.abs(xs / f - cc)
    ker = np.outer(k1, k1).astype(F32)
    return np.broadcast_to(ker, shape).copy()


# --------------------------------------------------------------------------
# backward ops (Caffe *_layer.cpp Backward_cpu restated)
# --------------------------------------------------------------------------

def col2im(col: np.ndarray, c: int, h: int, w: int, kh: int, kw: int, ph: int, pw: int, sh: int, sw: int) -> np.ndarray:
    """(C*kh*kw, OH*OW) -> (C,H,W): scatter-add, inverse of :func:`im2col`."""
    oh, ow = conv_out(h, kh, ph, sh), conv_out(w, kw, pw, sw)
    cols = col.reshape(c, kh, kw, oh, ow)
    xp = np.zeros((c, h + 2 * ph, w + 2 * pw), dtype=F32)
    for r in range(kh):
        for q in range(kw):
            xp[:, r:r + sh * oh:sh, q:q + sw * ow:sw] += cols[:, r, q]
    return xp[:, ph:ph + h, pw:pw + w]


def conv2d_backward(x: np.ndarray, w: np.ndarray, dy: np.ndarray, pad: int, stride: int, need_dx: bool = True):
    """Caffe ConvolutionLayer::Backward: dW = dY @ col^T summed over images, db = sum dY, dX = col2im(W^T @ dY)."""
    n, cin, h, wd = x.shape
    cout, _, kh, kw = w.shape
    dw = np.zeros_like(w)
    db = dy.sum(axis=(0, 2, 3)).astype(F32)
    dx = np.zeros_like(x) if need_dx else None
    wg = w.reshape(cout, cin * kh * kw)
    for i in range(n):
        col = im2col(x[i], kh, kw, pad, pad, stride, stride)
        dyi = dy[i].reshape(cout, -1)
        dw += (dyi @ col.T).reshape(w.shape)
        if need_dx:
            dx[i] = col2im(wg.T @ dyi, cin, h, wd, kh, kw, pad, pad, stride, stride)
    return dw, db, dx


def max_pool_backward(dy: np.ndarray, idx: np.ndarray, in_shape) -> np.ndarray:
    n, c, h, w = in_shape
    dx = np.zeros((n, c, h * w), dtype=F32)
    flat_idx = idx.reshape(n, c, -1)
    flat_dy = dy.reshape(n, c, -1)
    for i in range(n):
        for j in range(c):
            np.add.at(dx[i, j], flat_idx[i, j], flat_dy[i, j])
    return dx.reshape(n, c, h, w)


def lrn_across_backward(x: np.ndarray, y: np.ndarray, scale: np.ndarray, dy: np.ndarray, local_size: int, alpha: float,
                        beta: float) -> np.ndarray:
    """Caffe LRNLayer::CrossChannelBackward_cpu: dX = dY*scale^-beta - (2 alpha beta / n) * X * sum_window(dY*Y/scale)."""
    n, c, h, w = x.shape
    pre = (local_size - 1) // 2
    ratio = np.zeros((n, c + local_size - 1, h, w), dtype=F32)
    ratio[:, pre:pre + c] = dy * y / scale
    acc = np.zeros_like(x)
    for j in range(local_size):
        acc += ratio[:, j:j + c]
    return (dy * np.power(scale, F32(-beta)) - F32(2.0 * alpha * beta / local_size) * x * acc).astype(F32)


def sigmoid_backward(y: np.ndarray, dy: np.ndarray) -> np.ndarray:
    return (dy * y * (F32(1) - y)).astype(F32)


# counter-based dropout mask shared bit-for-bit with the HIP kernel (csrc/train.hip: dropout_keep):
# keep element i iff hash32(i ^ seed-mix) >= ratio * 2^32
def dropout_hash(index: np.ndarray, seed: int) -> np.ndarray:
    x = (index.astype(np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x.astype(np.uint64) * np.uint64(0x7FEB352D) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x.astype(np.uint64) * np.uint64(0x846CA68B) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def dropout_mask(shape, ratio: float, seed: int) -> np.ndarray:
    """1.0 where kept.  Index = NCHW linear index of the element."""
    idx = np.arange(int(np.prod(shape)), dtype=np.uint64)
    thresh = np.uint32(min(int(ratio * 4294967296.0), 4294967295))
    return (dropout_hash(idx, seed) >= thresh).astype(F32).reshape(shape)


# --------------------------------------------------------------------------
# solver updates (Caffe SGDSolver / AdamSolver ::ComputeUpdateValue + Regularize), one parameter blob
# --------------------------------------------------------------------------

def sgd_update(w, g, hist, lr, momentum, weight_decay, lr_mult, decay_mult):
    g = g + F32(weight_decay * decay_mult) * w
    hist[...] = F32(momentum) * hist + F32(lr * lr_mult) * g
    w -= hist


def adam_update(w, g, m, v, lr, beta1, beta2, delta, weight_decay, lr_mult, decay_mult, t):
    g = g + F32(weight_decay * decay_mult) * w
    m[...] = F32(beta1) * m + F32(1 - beta1) * g
    v[...] = F32(beta2) * v + F32(1 - beta2) * g * g
    corr = math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    w -= F32(lr * lr_mult * corr) * m / (np.sqrt(v) + F32(delta))
