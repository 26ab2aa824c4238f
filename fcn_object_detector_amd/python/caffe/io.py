"""caffe.io subset used by the reference (fcn_object_detector.py:319-322): Transformer.

The reference constructs a Transformer and calls its setters but never runs
``preprocess`` on the hot path; ``preprocess`` is provided for completeness and
follows pycaffe's documented order: resize -> transpose -> channel swap ->
raw scale -> mean subtraction -> input scale.
"""
from __future__ import annotations

import numpy as np


class Transformer(object):
    def __init__(self, inputs):
        self.inputs = dict(inputs)
        self.transpose = {}
        self.channel_swap = {}
        self.raw_scale = {}
        self.mean = {}
        self.input_scale = {}

    def _check(self, in_):
        if in_ not in self.inputs:
            raise Exception("{} is not one of the net inputs: {}".format(in_, list(self.inputs)))

    def set_transpose(self, in_, order):
        self._check(in_)
        if len(order) != len(self.inputs[in_]) - 1:
            raise Exception("Transpose order needs to have the same number of dimensions as the input.")
        self.transpose[in_] = tuple(order)

    def set_channel_swap(self, in_, order):
        self._check(in_)
        if len(order) != self.inputs[in_][1]:
            raise Exception("Channel swap needs to have the same number of dimensions as the input channels.")
        self.channel_swap[in_] = tuple(order)

    def set_raw_scale(self, in_, scale):
        self._check(in_)
        self.raw_scale[in_] = scale

    def set_input_scale(self, in_, scale):
        self._check(in_)
        self.input_scale[in_] = scale

    def set_mean(self, in_, mean):
        self._check(in_)
        ms = np.asarray(mean, dtype=np.float32)
        if ms.ndim == 1:
            if ms.shape[0] != self.inputs[in_][1]:
                raise ValueError("Mean channels incompatible with input.")
            ms = ms[:, np.newaxis, np.newaxis]
        self.mean[in_] = ms

    def preprocess(self, in_, data):
        self._check(in_)
        x = np.asarray(data, dtype=np.float32)
        in_dims = self.inputs[in_][2:]
        if x.shape[:2] != tuple(in_dims):
            x = _resize_bilinear(x, in_dims)
        if in_ in self.transpose:
            x = x.transpose(self.transpose[in_])
        if in_ in self.channel_swap:
            x = x[list(self.channel_swap[in_]), :, :]
        if in_ in self.raw_scale:
            x = x * self.raw_scale[in_]
        if in_ in self.mean:
            x = x - self.mean[in_]
        if in_ in self.input_scale:
            x = x * self.input_scale[in_]
        return x

    def deprocess(self, in_, data):
        self._check(in_)
        x = np.array(data, dtype=np.float32).squeeze()
        if in_ in self.input_scale:
            x = x / self.input_scale[in_]
        if in_ in self.mean:
            x = x + self.mean[in_]
        if in_ in self.raw_scale:
            x = x / self.raw_scale[in_]
        if in_ in self.channel_swap:
            x = x[np.argsort(self.channel_swap[in_]), :, :]
        if in_ in self.transpose:
            x = x.transpose(np.argsort(self.transpose[in_]))
        return x


def _resize_bilinear(img: np.ndarray, dims) -> np.ndarray:
    h, w = img.shape[:2]
    oh, ow = int(dims[0]), int(dims[1])
    ys = (np.arange(oh) + 0.5) * h / oh - 0.5
    xs = (np.arange(ow) + 0.5) * w / ow - 0.5
    y0 = np.clip(np.floor(ys).astype(int), 0, h - 1)
    x0 = np.clip(np.floor(xs).astype(int), 0, w - 1)
    y1 = np.clip(y0 + 1, 0, h - 1)
    x1 = np.clip(x0 + 1, 0, w - 1)
    fy = np.clip(ys - y0, 0, 1)[:, None, None]
    fx = np.clip(xs - x0, 0, 1)[None, :, None]
    im = img if img.ndim == 3 else img[:, :, None]
    out = (im[y0][:, x0] * (1 - fy) * (1 - fx) + im[y0][:, x1] * (1 - fy) * fx +
           im[y1][:, x0] * fy * (1 - fx) + im[y1][:, x1] * fy * fx)
    return out.astype(np.float32) if img.ndim == 3 else out[:, :, 0].astype(np.float32)
