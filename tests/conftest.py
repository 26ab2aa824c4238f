import os
import sys

# the oracle runs OpenBLAS (numpy) and OpenMP (oracle/caffe_cpu.c) side by side: keep their idle workers from spinning on each
# other's cores (both settings are read when the libraries load, so before numpy is imported)
os.environ.setdefault("OPENBLAS_THREAD_TIMEOUT", "4")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import numpy as np  # noqa: E402
import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("FCN_QUIET", "1")      # (lib.load() announces on stderr when it fills in GPU_MAX_HW_QUEUES)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PYCAFFE = os.path.join(ROOT, "fcn_object_detector_amd", "python")
GOLDEN = os.path.join(ROOT, "tests", "golden")
MODELS = os.path.join(ROOT, "tests", "golden", "nets")
# convolution configuration numbers (csrc/conv_fwd.hip): 0 .. N_TILE_CFGS-1 the tiled implicit-GEMM family, then the first-layer kernel,
# the lane-split 1x1 kernel, then the persistent half-float streaming kernel's.  tests/test_lib_abi.py holds them to the library.
N_TILE_CFGS = 30
CFG_FIRST7, CFG_DOT1X1, CFG_STREAM0 = N_TILE_CFGS, N_TILE_CFGS + 1, N_TILE_CFGS + 2


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available() -> bool:
    try:
        from fcn_object_detector_amd import lib as L
        import ctypes as C
        n = C.c_int(0)
        if L.load().fcn_device_count(C.byref(n)) != 0:
            return False
        return n.value > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests fail loudly (not skip) when selected with -m gpu but the HIP library/device is missing."""
    from fcn_object_detector_amd import lib as L
    L.load()
    if not _gpu_available():
        pytest.fail("no HIP device visible — -m gpu tests must run on the MI355X box")
    L.call("fcn_init", 0)
    return True


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def elem_err(a, b, tol=1e-3):
    """Element-wise criterion beside rel_err: every element must satisfy
    |a - b| <= tol * |b| + tol * rms(b).  rel_err normalises by the blob's LARGEST value, so in a blob that spans orders of
    magnitude (`bboxes`, parameter gradients) a small element may be off by far more than `tol` of itself and still pass;
    here the allowance of an element is `tol` of its own magnitude plus `tol` of the blob's typical magnitude (the rms term
    covers sums that cancel to ~0).  Returns (worst ratio |a-b| / allowance, flat index of the worst element): <= 1 passes."""
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    if a.size == 0:
        return 0.0, -1
    allow = tol * np.abs(b) + tol * max(float(np.sqrt(np.mean(b * b))), 1e-30)
    r = np.abs(a - b) / allow
    i = int(np.argmax(r))
    return float(r[i]), i
