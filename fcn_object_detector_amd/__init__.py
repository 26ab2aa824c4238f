"""fcn_object_detector_amd — MI355X (gfx950) engine for the fcn_object_detector hot path.

Layout:
  csrc/            hand-written HIP kernels + the C ABI (include/fcnhip.h) -> libfcnhip.so
  lib.py           ctypes binding (fails loudly when the library is missing; no CPU fallback)
  proto.py         prototxt / caffemodel codecs
  netspec.py       layer graph, shape rules, fillers
  engine.py        launch plan + hipGraph executor
  detector.py      host mirror of the reference's inference node (pre/post-processing on device)
  python/caffe/    pycaffe-compatible front end (``import caffe``)
"""
__all__ = ["lib", "proto", "netspec", "engine", "detector"]
