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
