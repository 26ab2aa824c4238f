"""pycaffe-compatible front end over the MI355X engine (fcn_object_detector_amd).

Put ``<repo>/fcn_object_detector_amd/python`` on PYTHONPATH — the same way the
reference puts ``$CAFFE_ROOT/python`` there (reference: train/train.sh:19-22) —
and ``import caffe`` resolves to this package.  Only the surface the reference's
hot path uses is provided:

  caffe.Net(proto, weights, caffe.TEST)          fcn_object_detector.py:317
  net.blobs[name].data / .reshape(...)            fcn_object_detector.py:82,89-90,324-328
  net.forward()                                   fcn_object_detector.py:87
  caffe.set_device / caffe.set_mode_gpu           fcn_object_detector.py:68-69
  caffe.io.Transformer(...)                       fcn_object_detector.py:319-322
  caffe.Layer (setup/reshape/forward/backward)    data_argumentation_layer.py:14-127

All arithmetic runs in libfcnhip.so on the GPU; there is no CPU mode.
"""
from __future__ import annotations

import os
import sys
import threading
from collections import OrderedDict
from typing import Dict, List, Optional

import numpy as np

_PKG_PARENT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _PKG_PARENT not in sys.path:
    sys.path.insert(0, _PKG_PARENT)

from fcn_object_detector_amd import lib as _L  # noqa: E402
from fcn_object_detector_amd import proto as _proto  # noqa: E402
from fcn_object_detector_amd import pylayer as _pylayer  # noqa: E402
from fcn_object_detector_amd.engine import Engine as _Engine  # noqa: E402
from fcn_object_detector_amd.netspec import NetSpec as _NetSpec, fill_params as _fill_params  # noqa: E402

from . import io  # noqa: E402,F401

TRAIN = 0
TEST = 1
__version__ = "1.0.0-fcnhip"

_state = threading.local()
_default_device = 0


def set_device(device_id: int) -> None:
    """Select the GPU for the calling thread (Caffe's mode/device are thread-local as well)."""
    global _default_device
    _state.device = int(device_id)
    _default_device = int(device_id)
    _L.call("fcn_init", int(device_id))


def set_mode_gpu() -> None:
    _L.call("fcn_init", getattr(_state, "device", _default_device))


def set_mode_cpu() -> None:
    raise RuntimeError("this caffe front end has no CPU mode: the engine is libfcnhip.so on an MI355X")


def _current_device() -> int:
    return getattr(_state, "device", _default_device)


Layer = _pylayer.Layer
_TopProxy = _pylayer.TopProxy


class _Blob(object):
    """``net.blobs[name]``: ``.data`` is a writable NCHW float32 array synchronised with the device."""

    def __init__(self, net: "Net", name: str):
        self._net = net
        self._name = name

    @property
    def data(self) -> np.ndarray:
        return self._net._blob_data(self._name)

    @property
    def shape(self):
        return tuple(self._net._shape(self._name))

    @property
    def num(self):
        return self.shape[0]

    @property
    def channels(self):
        return self.shape[1]

    @property
    def height(self):
        return self.shape[2]

    @property
    def width(self):
        return self.shape[3]

    @property
    def count(self):
        return int(np.prod(self.shape)) if self.shape else 1

    def reshape(self, *dims) -> None:
        dims = tuple(int(d) for d in (dims[0] if len(dims) == 1 and isinstance(dims[0], (tuple, list)) else dims))
        self._net._reshape_blob(self._name, dims)


class _Param(object):
    """``net.params[layer][i]``: Caffe-layout (OIHW / bias) host array; edits are uploaded at the next forward."""

    def __init__(self, net: "Net", layer: str, index: int):
        self._net, self._layer, self._index = net, layer, index

    @property
    def data(self) -> np.ndarray:
        self._net._params_touched.add(self._layer)
        return self._net._engine.params_host[self._layer][self._index]

    @property
    def shape(self):
        return self._net._engine.params_host[self._layer][self._index].shape


class Net(object):
    def __init__(self, network_file, *args, **kwargs):
        weights = kwargs.pop("weights", None)
        phase = kwargs.pop("phase", None)
        # extension (not in pycaffe): dtype="f16" / $FCN_DTYPE=f16 runs a TEST net with half-float activations and weights
        self._dtype = str(kwargs.pop("dtype", os.environ.get("FCN_DTYPE", "f32")))
        for a in args:
            if isinstance(a, (int, np.integer)) and not isinstance(a, bool):
                phase = int(a)
            elif a is not None:
                weights = a
        if kwargs:
            raise TypeError("unexpected arguments to caffe.Net: %s" % sorted(kwargs))
        if phase is None:
            raise TypeError("caffe.Net needs a phase (caffe.TRAIN or caffe.TEST)")
        if not os.path.isfile(str(network_file)):
            raise IOError("network file not found: %s" % network_file)
        if weights is not None and not os.path.isfile(str(weights)):
            raise IOError("weights file not found: %s" % weights)
        self._phase = "TRAIN" if phase == TRAIN else "TEST"
        self._proto_path = str(network_file)
        self._msg = _proto.parse_file(self._proto_path)
        self._device = _current_device()
        self._lock = threading.RLock()
        self._py_layers: List[tuple] = []
        self._data_shapes: Dict[str, tuple] = {}
        self._params_touched = set()
        self._user_shapes: Dict[str, tuple] = {}
        self._engine: Optional[_Engine] = None
        self._setup_python_layers()
        self._build(initial_params=None)
        if weights is not None:
            self.copy_from(str(weights))

    # ---- construction -------------------------------------------------
    def _setup_python_layers(self) -> None:
        spec = _NetSpec(self._msg, self._phase)
        self._py_layers, self._data_shapes = _pylayer.setup_python_layers(spec, TRAIN if self._phase == "TRAIN" else TEST)

    def _build(self, initial_params) -> None:
        spec = _NetSpec(self._msg, self._phase)
        shapes = dict(self._data_shapes)
        shapes.update(self._user_shapes)
        params = initial_params
        if params is None:
            spec.infer({**spec.input_shapes, **shapes})
            params = _fill_params(spec, seed=0)
        self._spec = spec
        self._engine = _Engine(spec, data_shapes=shapes, params=params, device=self._device,
                               dtype=self._dtype if self._phase == "TEST" else "f32")
        self.blobs = OrderedDict((name, _Blob(self, name)) for name in self._engine.shapes)
        self.params = OrderedDict(
            (l.name, [_Param(self, l.name, i) for i in range(len(self._engine.params_host[l.name]))])
            for l in spec.param_layers())
        self.inputs = list(spec.input_shapes.keys())
        self.outputs = list(self._engine.outputs)
        self._touched_inputs = set(self._engine.inputs)
        for l, inst, bottoms, tops in self._py_layers:
            # a layer that renders on the device (the bundled data layer) writes its image tops straight into HBM
            if getattr(inst, "supports_device_scenes", False):
                inst.bind_device(self._engine, [t.name for t in tops])

    def _shape(self, name: str):
        return self._engine.shapes[name]

    def _reshape_blob(self, name: str, dims) -> None:
        with self._lock:
            if tuple(self._engine.shapes[name]) == tuple(dims):
                return
            if name not in self._engine.inputs:
                raise ValueError("only input blobs can be reshaped (blob %r)" % name)
            self._user_shapes[name] = tuple(dims)
            params = {k: [a.copy() for a in v] for k, v in self._engine.params_host.items()}
            self._engine.close()
            self._build(initial_params=params)

    def _blob_data(self, name: str) -> np.ndarray:
        eng = self._engine
        if name in eng.inputs:
            self._touched_inputs.add(name)
            return eng.host_array(name)
        return eng.read_blob(name)

    # ---- weights ------------------------------------------------------
    def copy_from(self, weights_path: str) -> None:
        """Load a binary .caffemodel by layer name (Net::CopyTrainedLayersFrom)."""
        if not os.path.isfile(weights_path):
            raise IOError("weights file not found: %s" % weights_path)
        _proto.copy_trained_layers(weights_path, self._engine.params_host, self._engine.set_params)

    def save(self, path: str) -> None:
        layers = [(l.name, l.type, self._engine.params_host[l.name]) for l in self._spec.param_layers()]
        _proto.write_caffemodel(path, layers, self._spec.name)

    # ---- execution ----------------------------------------------------
    def forward(self, blobs=None, start=None, end=None, **kwargs) -> Dict[str, np.ndarray]:
        if start is not None or end is not None:
            raise NotImplementedError("partial forward (start/end) is not used by the reference")
        with self._lock:
            _L.call("fcn_init", self._device)
            eng = self._engine
            for lname in list(self._params_touched):
                eng.set_params(lname, eng.params_host[lname])
            self._params_touched.clear()
            for k, v in kwargs.items():
                if k not in eng.inputs:
                    raise KeyError("forward(): %r is not an input blob" % k)
                eng.host_array(k)[...] = v
            for l, inst, bottoms, tops in self._py_layers:
                inst.reshape(bottoms, tops)
                inst.forward(bottoms, tops)
                for t in tops:
                    if tuple(t.shape_) != tuple(eng.shapes[t.name]):
                        raise NotImplementedError("Python layer %s changed the shape of %s" % (l.name, t.name))
                    if t.name in getattr(inst, "device_tops", ()):
                        eng.device_fed.add(t.name)
                    else:
                        eng.host_array(t.name)[...] = t.data
            out = eng.forward()
            res = {k: out[k] for k in self.outputs}
            if blobs:
                for b in blobs:
                    res[b] = eng.read_blob(b)
            return res

    def score_masks(self, rects, frame_hw, prob_thresh, score_blob="score", padding=10):
        """Extension (not in pycaffe): what run_detector2 does with net.blobs['score'].data after net.forward() (reference:
        scripts/fcn_object_detector.py:208-236 with create_mask_labels :279-303), on the device: -> (pmap (h, w) uint8,
        [(np.array([x, y, w, h]), class index), ...]).  `rects`: the windows' (x, y, w, h) in the frame, one per image of the batch."""
        from fcn_object_detector_amd.detector import ScoreMasks
        with self._lock:
            _L.call("fcn_init", self._device)
            eng = self._engine
            sb = eng.blobs[score_blob]
            if sb.esize != 4 or len(sb.shape) != 4:
                raise NotImplementedError("score blob %s must be a 4-d float32 blob" % score_blob)
            rects = np.ascontiguousarray(rects, np.int32).reshape(-1, 4)
            n, c, sh, sw = sb.shape
            if len(rects) != n:
                raise ValueError("%d windows for a batch of %d" % (len(rects), n))
            key = (n, c, int(rects[0, 2]), int(rects[0, 3]), int(frame_hw[0]), int(frame_hw[1]))
            if getattr(self, "_score_masks_key", None) != key:
                self._score_masks, self._score_masks_key = ScoreMasks(*key), key
            for op in eng._lazy_blob_ops.get(score_blob, ()):
                op.run(eng.stream)
            self._score_masks.launch(sb.buf.ptr, sh, sw, sb.cstride, sb.coffset, rects, prob_thresh, eng.stream)
            return self._score_masks.fetch(eng.stream, padding)

    def backward(self, **kwargs):
        raise NotImplementedError("Net.backward(): training runs through the `caffe train` tool")


# ---- solvers (pycaffe: caffe.SGDSolver / caffe.AdamSolver / caffe.get_solver) --------------------------------------
def get_solver(solver_file, **kw):
    from fcn_object_detector_amd.solver import Solver
    kw.setdefault("device", _current_device())
    return Solver(str(solver_file), **kw)


SGDSolver = get_solver
AdamSolver = get_solver
