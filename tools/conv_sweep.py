#!/usr/bin/env python
"""GPU micro-benchmark: one convolution problem (or a group) across all tile configurations.
usage: python tools/conv_sweep.py  [--shapes name]   (run on the GPU box)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fcn_object_detector_amd import lib as L  # noqa: E402
from fcn_object_detector_amd.engine import DeviceBuffer  # noqa: E402
from gpu_util import conv_desc, dev_from  # noqa: E402

# (name, [(Cin, Cout, k, pad, stride, H, W), ...])  problems sharing one launch
SHAPES = [
    ("conv1", [(4, 64, 7, 3, 2, 448, 448)]),
    ("conv2_3x3", [(64, 192, 3, 1, 1, 112, 112)]),
    ("3a_A", [(192, 64, 1, 0, 1, 56, 56), (192, 96, 1, 0, 1, 56, 56), (192, 16, 1, 0, 1, 56, 56)]),
    ("3b_B", [(128, 192, 3, 1, 1, 56, 56), (32, 96, 5, 2, 1, 56, 56), (256, 64, 1, 0, 1, 56, 56)]),
    ("4a_A", [(480, 192, 1, 0, 1, 28, 28), (480, 96, 1, 0, 1, 28, 28), (480, 16, 1, 0, 1, 28, 28)]),
    ("4a_B", [(96, 208, 3, 1, 1, 28, 28), (16, 48, 5, 2, 1, 28, 28), (480, 64, 1, 0, 1, 28, 28)]),
    ("4e_3x3", [(160, 320, 3, 1, 1, 28, 28)]),
    ("5b_A", [(832, 384, 1, 0, 1, 28, 28), (832, 192, 1, 0, 1, 28, 28), (832, 48, 1, 0, 1, 28, 28)]),
    ("5b_B", [(192, 384, 3, 1, 1, 28, 28), (48, 128, 5, 2, 1, 28, 28), (832, 128, 1, 0, 1, 28, 28)]),
    ("heads", [(1024, 4, 1, 0, 1, 28, 28), (1024, 16, 1, 0, 1, 28, 28)]),
    ("conv2_red", [(64, 64, 1, 0, 1, 112, 112)]),
    ("3a_B", [(96, 128, 3, 1, 1, 56, 56), (16, 32, 5, 2, 1, 56, 56), (192, 32, 1, 0, 1, 56, 56)]),
    ("3b_A", [(256, 128, 1, 0, 1, 56, 56), (256, 128, 1, 0, 1, 56, 56), (256, 32, 1, 0, 1, 56, 56)]),
    ("3b_3x3", [(128, 192, 3, 1, 1, 56, 56)]),
    ("4c_3x3", [(128, 256, 3, 1, 1, 28, 28)]),
    ("4c_B", [(128, 256, 3, 1, 1, 28, 28), (24, 64, 5, 2, 1, 28, 28), (512, 64, 1, 0, 1, 28, 28)]),
    ("5b_3x3", [(192, 384, 3, 1, 1, 28, 28)]),
    # K scaling at M=784, N=320 (3x3): separates the fixed cost from the per-chunk cost
    ("k288", [(32, 320, 3, 1, 1, 28, 28)]),
    ("k576", [(64, 320, 3, 1, 1, 28, 28)]),
    ("k1440", [(160, 320, 3, 1, 1, 28, 28)]),
    ("k2880", [(320, 320, 3, 1, 1, 28, 28)]),
    ("k5760", [(640, 320, 3, 1, 1, 28, 28)]),
    # same K, 1x1 (no halo / tap logic): K = Cin
    ("p1440", [(1440, 320, 1, 0, 1, 28, 28)]),
    # round 3: what the 5x5 convolutions on 16 / 32 channels cost inside their launches (a 64-channel chunk per tap is 75 / 50 % padding)
    ("3a_5x5", [(16, 32, 5, 2, 1, 56, 56)]),
    ("3a_B_no5", [(96, 128, 3, 1, 1, 56, 56), (192, 32, 1, 0, 1, 56, 56)]),
    ("3b_5x5", [(32, 96, 5, 2, 1, 56, 56)]),
    ("3b_B_no5", [(128, 192, 3, 1, 1, 56, 56), (256, 64, 1, 0, 1, 56, 56)]),
    ("4a_5x5", [(16, 48, 5, 2, 1, 28, 28)]),
    ("4a_B_no5", [(96, 208, 3, 1, 1, 28, 28), (480, 64, 1, 0, 1, 28, 28)]),
    ("4e_B", [(160, 320, 3, 1, 1, 28, 28), (32, 128, 5, 2, 1, 28, 28)]),
    ("4e_5x5", [(32, 128, 5, 2, 1, 28, 28)]),
    # round 4: what would ONE launch for a module's 3x3 level and the next module's reduce level cost, dependencies aside (an upper bound
    # on what sharing a launch can save: no waiting is modelled)
    ("4b_A", [(512, 160, 1, 0, 1, 28, 28), (512, 112, 1, 0, 1, 28, 28), (512, 24, 1, 0, 1, 28, 28)]),
    ("4aB+4bA", [(96, 208, 3, 1, 1, 28, 28), (16, 48, 5, 2, 1, 28, 28), (480, 64, 1, 0, 1, 28, 28),
                 (512, 160, 1, 0, 1, 28, 28), (512, 112, 1, 0, 1, 28, 28), (512, 24, 1, 0, 1, 28, 28)]),
    ("5a_A", [(832, 256, 1, 0, 1, 28, 28), (832, 160, 1, 0, 1, 28, 28), (832, 32, 1, 0, 1, 28, 28)]),
    ("4eB+5aA", [(160, 320, 3, 1, 1, 28, 28), (32, 128, 5, 2, 1, 28, 28), (528, 128, 1, 0, 1, 28, 28),
                 (832, 256, 1, 0, 1, 28, 28), (832, 160, 1, 0, 1, 28, 28), (832, 32, 1, 0, 1, 28, 28)]),
    ("4e_Bfull", [(160, 320, 3, 1, 1, 28, 28), (32, 128, 5, 2, 1, 28, 28), (528, 128, 1, 0, 1, 28, 28)]),
    # round 4 (half floats, batch 32): what pool_proj costs inside its level, and alone - the budget of a fused pooling + pool_proj kernel
    ("3a_B_nopp", [(96, 128, 3, 1, 1, 56, 56), (16, 32, 5, 2, 1, 56, 56)]),
    ("3a_pp", [(192, 32, 1, 0, 1, 56, 56)]),
    ("4a_B_nopp", [(96, 208, 3, 1, 1, 28, 28), (16, 48, 5, 2, 1, 28, 28)]),
    ("4a_pp", [(480, 64, 1, 0, 1, 28, 28)]),
    ("5b_B_nopp", [(192, 384, 3, 1, 1, 28, 28), (48, 128, 5, 2, 1, 28, 28)]),
    ("5b_pp", [(832, 128, 1, 0, 1, 28, 28)]),
]


CFGS = [int(c) for c in os.environ.get("SWEEP_CFGS", "").split(",") if c]
F16 = bool(os.environ.get("SWEEP_F16"))          # half-float activations / weights (v_mfma_f32_32x32x16_f16)
BATCH = int(os.environ.get("SWEEP_BATCH", "1"))


def main():
    want = sys.argv[1:] or None
    L.call("fcn_init", 0)
    lib = L.load()
    sp = C.c_void_p()
    L.call("fcn_stream_create", C.byref(sp))
    st = sp.value
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.call("fcn_event_create", C.byref(e0))
    L.call("fcn_event_create", C.byref(e1))
    rng = np.random.default_rng(0)
    for name, probs in SHAPES:
        if want and name not in want:
            continue
        keep, descs, flops = [], [], 0.0
        for (cin, cout, k, pad, s, h, w) in probs:
            oh, ow = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
            dt = np.float16 if F16 else np.float32
            if F16:
                cin = (cin + 7) // 8 * 8
            co = (cout + 7) // 8 * 8 if F16 else cout
            x = dev_from(rng.standard_normal((BATCH, h, w, cin)).astype(dt))
            wt = dev_from((rng.standard_normal((cout, k, k, cin)) * 0.05).astype(dt))
            b = dev_from(np.zeros(cout, np.float32))
            y = dev_from(np.zeros((BATCH, oh, ow, co), dt))
            keep += [x, wt, b, y]
            descs.append(conv_desc(x, wt, b, y, BATCH, h, w, cin, cin, cout, k, pad, s, oh, ow, co, 0, L.CONV_RELU | (L.CONV_F16 if F16 else 0)))
            flops += 2.0 * BATCH * oh * ow * cout * cin * k * k
        arr = (L.ConvDesc * len(descs))(*descs)
        ws = DeviceBuffer(int(lib.fcn_conv2d_group_workspace_bytes(len(descs))), zero=False)
        line = "%-10s %6.3f GFLOP |" % (name, flops / 1e9)
        for cfg in ["auto"] + [str(i) for i in (CFGS or range(lib.fcn_conv2d_num_configs()))]:
            if cfg == "auto":
                os.environ.pop("FCN_CONV_CFG", None)
            else:
                os.environ["FCN_CONV_CFG"] = cfg
            grp = L.ConvGroup()
            if lib.fcn_conv2d_group_prepare(arr, len(descs), ws.ptr, -1 if cfg == "auto" else int(cfg), C.byref(grp)) != 0:
                continue      # (the first-layer kernel only takes conv1-shaped problems)
            for _ in range(3):
                L.call("fcn_conv2d_fwd_group_f32", C.byref(grp), st)
            L.call("fcn_event_record", e0, st)
            reps = 30
            for _ in range(reps):
                L.call("fcn_conv2d_fwd_group_f32", C.byref(grp), st)
            L.call("fcn_event_record", e1, st)
            L.call("fcn_event_sync", e1)
            ms = C.c_float()
            L.call("fcn_event_elapsed_ms", e0, e1, C.byref(ms))
            us = ms.value / reps * 1e3
            line += " %s:c%d/%dt %6.1fus %5.1fTF |" % (cfg if cfg != "auto" else "A", grp.cfg, grp.total_tiles, us, flops / us / 1e6)
        print(line, flush=True)


if __name__ == "__main__":
    main()
