"""Host-side support for ``type: 'Python'`` layers (the reference's data layer is one:
scripts/data_argumentation_layer/data_argumentation_layer.py:14, wired up in README.md:57-76).

Shared by the pycaffe front end (``caffe.Net``) and the solver (``caffe train``): the layer object is created from
``python_param { module, layer, param_str }``, sees proxies with ``reshape(*dims)`` / ``.data`` as its tops, and its
outputs become input blobs of the device engine.
"""
from __future__ import annotations

import importlib
from typing import Dict, List, Tuple

import numpy as np

TRAIN = 0
TEST = 1


class Layer(object):
    """Base class of Python layers; mirrors caffe.Layer."""

    def __init__(self):
        self.param_str = ""
        self.blobs = []
        self.phase = TEST

    def setup(self, bottom, top):
        pass

    def reshape(self, bottom, top):
        pass

    def forward(self, bottom, top):
        pass

    def backward(self, top, propagate_down, bottom):
        pass


class TopProxy(object):
    """What a Python layer sees as ``top[i]`` / ``bottom[i]``: ``reshape(*dims)`` and a float32 ``data`` array."""

    def __init__(self, name: str):
        self.name = name
        self.shape_ = None
        self._data = None
        self.diff = None

    def reshape(self, *dims):
        dims = tuple(int(d) for d in (dims[0] if len(dims) == 1 and isinstance(dims[0], (tuple, list)) else dims))
        if self.shape_ != dims:
            self.shape_ = dims
            if self._data is None or self._data.shape != dims:
                self._data = np.zeros(dims, np.float32)

    @property
    def data(self):
        return self._data

    @property
    def shape(self):
        return self.shape_

    @property
    def num(self):
        return self.shape_[0]

    @property
    def channels(self):
        return self.shape_[1]

    @property
    def height(self):
        return self.shape_[2]

    @property
    def width(self):
        return self.shape_[3]


def setup_python_layers(spec, phase: int) -> Tuple[List[tuple], Dict[str, tuple]]:
    """Instantiate every Python layer of ``spec``; returns [(layer, instance, bottoms, tops)] and {top blob: shape}."""
    out, shapes = [], {}
    for l in spec.layers:
        if l.type != "Python":
            continue
        pp = l.sub("python_param")
        mod = importlib.import_module(str(pp.get("module")))
        cls = getattr(mod, str(pp.get("layer")))
        inst = cls.__new__(cls)
        Layer.__init__(inst)
        try:
            cls.__init__(inst)
        except TypeError:
            pass
        inst.param_str = str(pp.get("param_str", ""))
        inst.phase = phase
        bottoms = [TopProxy(b) for b in l.bottoms]
        tops = [TopProxy(t) for t in l.tops]
        inst.setup(bottoms, tops)
        inst.reshape(bottoms, tops)
        for t in tops:
            if t.shape_ is None:
                raise RuntimeError("Python layer %s did not reshape top %s" % (l.name, t.name))
            shapes[t.name] = t.shape_
        out.append((l, inst, bottoms, tops))
    return out, shapes
