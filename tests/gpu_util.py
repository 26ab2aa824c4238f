"""Helpers for the GPU parity tests: move NCHW numpy arrays to NHWC device buffers and back through the C ABI."""
import ctypes as C

import numpy as np

from fcn_object_detector_amd import lib as L
from fcn_object_detector_amd.engine import DeviceBuffer


def dev_from(arr: np.ndarray) -> DeviceBuffer:
    a = np.ascontiguousarray(arr)
    d = DeviceBuffer(max(a.nbytes, 16), zero=False)
    L.call("fcn_memcpy_h2d_async", d.ptr, a.ctypes.data, a.nbytes, None)
    L.call("fcn_device_sync")
    return d


def dev_to(d: DeviceBuffer, shape, dtype=np.float32) -> np.ndarray:
    out = np.empty(shape, dtype)
    L.call("fcn_memcpy_d2h_async", out.ctypes.data, d.ptr, out.nbytes, None)
    L.call("fcn_device_sync")
    return out


def nhwc(x: np.ndarray, cstride=None, coffset=0, fill=0.0) -> np.ndarray:
    """NCHW -> NHWC with optional wider channel stride / offset (pad filled with `fill`)."""
    n, c, h, w = x.shape
    cs = cstride or c
    out = np.full((n, h, w, cs), fill, np.float32)
    out[..., coffset:coffset + c] = x.transpose(0, 2, 3, 1)
    return out


def nchw(y: np.ndarray, c: int, coffset=0) -> np.ndarray:
    return np.ascontiguousarray(y[..., coffset:coffset + c].transpose(0, 3, 1, 2))


def conv_desc(x_dev, w_dev, b_dev, y_dev, N, H, W, Cin, x_cstride, Cout, k, pad, stride, OH, OW, y_cstride, y_coffset=0, flags=0,
              in_shift=0.0, y2_dev=None, y2_cstride=0, y2_coffset=0):
    d = L.ConvDesc()
    d.x, d.w, d.bias, d.y = x_dev.ptr, w_dev.ptr, (b_dev.ptr if b_dev is not None else None), y_dev.ptr
    d.y2 = y2_dev.ptr if y2_dev is not None else None
    d.N, d.H, d.W, d.Cin, d.x_cstride = N, H, W, Cin, x_cstride
    d.Cout, d.kh, d.kw, d.pad, d.stride, d.OH, d.OW = Cout, k, k, pad, stride, OH, OW
    d.y_cstride, d.y_coffset, d.y2_cstride, d.y2_coffset = y_cstride, y_coffset, y2_cstride, y2_coffset
    d.flags, d.in_shift = flags, in_shift
    return d


def pack_ohwi(w: np.ndarray) -> np.ndarray:
    co, ci, kh, kw = w.shape
    ci4 = (ci + 3) // 4 * 4
    out = np.zeros((co, kh, kw, ci4), np.float32)
    out[..., :ci] = w.transpose(0, 2, 3, 1)
    return out


def adopt_device_activations(ref, eng, spec, keep=()):
    """Make the oracle's backward run on the DEVICE's forward pass: every 4-d blob of `ref` (except `keep`, the inputs) is
    replaced by the engine's, and the pooling argmaxes / LRN scales are recomputed from them.  ReLU masks and max-pool
    argmaxes are discontinuous: two independently rounded forward passes flip a handful of near-zero activations / near-tied
    windows, and a flipped mask says nothing about the backward kernels or the solver.  With identical masks what is left
    is the backward arithmetic itself, which is held to the north-star tolerance (1e-3)."""
    from oracle import caffe_ref as R
    for name in list(ref.blobs):
        if name in eng.blobs and len(eng.blobs[name].shape) == 4 and name not in keep:
            ref.blobs[name] = eng.read_blob(name).copy()
    for l in spec.layers:
        if l.type == "Pooling" and str(l.sub("pooling_param").get("pool", "MAX")) == "MAX":
            k, s_, p_ = (int(l.sub("pooling_param").get(q, d)) for q, d in (("kernel_size", 0), ("stride", 1), ("pad", 0)))
            ref.aux[l.name] = R.max_pool(ref.blobs[l.bottoms[0]], k, s_, p_, return_index=True)[1]
        elif l.type == "LRN":
            ref.aux[l.name] = R.lrn_across(ref.blobs[l.bottoms[0]], 5, 1e-4, 0.75, 1.0, return_scale=True)[1]
