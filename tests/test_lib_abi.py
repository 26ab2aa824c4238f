"""The C-ABI library builds for gfx950, loads, and exports every symbol include/fcnhip.h declares (no GPU needed)."""
import ctypes
import os
import re

from conftest import ROOT
from fcn_object_detector_amd import lib as L


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "fcnhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fcn_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    if not os.path.isfile(L.LIB_PATH):
        g.build()
    lib = L.load()
    names = _declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), "libfcnhip.so lacks %s" % n
        assert n in L.PROTOTYPES, "lib.py has no prototype for %s" % n
    assert sorted(L.PROTOTYPES) == names
    assert lib.fcn_abi_version() == 1


def test_struct_layouts_match_header():
    # sizes implied by the C declarations (LP64): guards against drift between fcnhip.h and lib.py
    assert ctypes.sizeof(L.ConvDesc) == 5 * 8 + 18 * 4 and L.ConvDesc.N.offset == 40 and L.ConvDesc.in_shift.offset == 108
    assert ctypes.sizeof(L.ConvGroup) == 24
    assert L.DetectParams.eps.offset == 48 and ctypes.sizeof(L.DetectParams) == 72


def test_argument_validation_without_gpu():
    lib = L.load()
    d = L.ConvDesc()
    assert lib.fcn_conv2d_fwd_f32(ctypes.byref(d), None) == 1            # FCN_E_ARG: null pointers
    assert b"null" in lib.fcn_last_error_string()
    assert lib.fcn_conv2d_fwd_f32(None, None) == 1
    assert lib.fcn_maxpool_fwd_f32(None, None, None, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 0, None) == 1


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "fcn_object_detector_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, os.path.join(dp, f)


def test_configuration_numbers_the_tests_name_are_the_librarys():
    """tests/conftest.py names the convolution configurations (tile family, first-layer, lane-split 1x1, streaming): hold the names to the library."""
    from conftest import CFG_DOT1X1, CFG_FIRST7, CFG_STREAM0, N_TILE_CFGS
    from fcn_object_detector_amd import lib as L
    lib = L.load()
    assert int(lib.fcn_conv2d_first_layer_config()) == CFG_FIRST7 == N_TILE_CFGS and CFG_DOT1X1 == CFG_FIRST7 + 1
    assert int(lib.fcn_conv2d_num_configs()) == CFG_STREAM0 + 11
    assert int(lib.fcn_conv2d_config_waves_k(23)) == 4 and int(lib.fcn_conv2d_config_lds_bytes(23)) == 32 * 1024
    assert int(lib.fcn_conv2d_config_lds_bytes(CFG_STREAM0)) > 100 * 1024 and int(lib.fcn_conv2d_config_lds_bytes(CFG_STREAM0 + 11)) == -1
