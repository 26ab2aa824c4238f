#!/usr/bin/env python
"""GPU diagnostic: where the time of ONE convolution launch goes, from clock stamps taken inside the kernel.

Needs the stamped build (`make -C fcn_object_detector_amd/csrc stamps`) and loads it through $FCN_LIB_PATH:

    FCN_LIB_PATH=fcn_object_detector_amd/libfcnhip_stamps.so python tools/conv_timeline.py [shape ...]

Per shape and tile configuration: the launch's HIP-event time (back to back, caches warm) and, from one stamped launch,
the spread over workgroups of: entry, prologue issued, first chunk usable, main loop done, K-split reduction done, stores
issued, stores acknowledged - all in microseconds after the first workgroup's entry (constant 100 MHz clock)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("FCN_LIB_PATH", os.path.join(ROOT, "fcn_object_detector_amd", "libfcnhip_stamps.so"))
from fcn_object_detector_amd import lib as L  # noqa: E402
from fcn_object_detector_amd.engine import DeviceBuffer  # noqa: E402
from gpu_util import conv_desc, dev_from  # noqa: E402
from conv_sweep import SHAPES  # noqa: E402

NAMES = ("entry", "setup", "issued", "chunk0", "iter0", "iter1", "iter2", "iter3", "loop", "parked", "stored", "acked")
ORDER = (0, 7, 1, 2, 8, 9, 10, 11, 3, 4, 5, 6)      # stamp slot of each name
CAP = 4096
F16 = bool(os.environ.get("SWEEP_F16"))          # half-float problems (as tools/conv_sweep.py)
BATCH = int(os.environ.get("SWEEP_BATCH", "1"))


def main():
    want = sys.argv[1:] or ["4a_A", "4a_B", "5b_B", "heads", "conv2_3x3", "3a_A"]
    cfgs_env = [int(c) for c in os.environ.get("SWEEP_CFGS", "").split(",") if c]
    cold = bool(os.environ.get("TIMELINE_COLD"))
    L.call("fcn_init", 0)
    lib = L.load()
    lib.fcn_debug_conv_stamps.restype = C.c_int
    lib.fcn_debug_conv_stamps.argtypes = [C.c_void_p, C.c_int]
    sp = C.c_void_p()
    L.call("fcn_stream_create", C.byref(sp))
    st = sp.value
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.call("fcn_event_create", C.byref(e0))
    L.call("fcn_event_create", C.byref(e1))
    stamps = DeviceBuffer(CAP * 32 * 8)
    flush = DeviceBuffer(512 << 20, zero=False)
    rng = np.random.default_rng(0)
    for name, probs in SHAPES:
        if name not in want:
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
        timed = []
        for cfg in (cfgs_env or range(lib.fcn_conv2d_num_configs())):
            grp = L.ConvGroup()
            if lib.fcn_conv2d_group_prepare(arr, len(descs), ws.ptr, cfg, C.byref(grp)) != 0 or grp.total_tiles > CAP:
                continue
            L.check(lib.fcn_debug_conv_stamps(None, 0))
            for _ in range(3):
                L.call("fcn_conv2d_fwd_group_f32", C.byref(grp), st)
            L.call("fcn_event_record", e0, st)
            for _ in range(30):
                L.call("fcn_conv2d_fwd_group_f32", C.byref(grp), st)
            L.call("fcn_event_record", e1, st)
            L.call("fcn_event_sync", e1)
            ms = C.c_float()
            L.call("fcn_event_elapsed_ms", e0, e1, C.byref(ms))
            timed.append((ms.value / 30 * 1e3, cfg, grp.total_tiles))
        timed.sort()
        print("== %s  %.3f GFLOP  (ideal %.2f us at 157.3 TF)   best configs: %s" % (
            name, flops / 1e9, flops / 157.3e6, "  ".join("c%d/%dwg %.1fus" % (c, t, u) for u, c, t in timed[:6])), flush=True)
        for us, cfg, tiles in timed[:int(os.environ.get("TIMELINE_TOP", "3"))]:
            grp = L.ConvGroup()
            L.call("fcn_conv2d_group_prepare", arr, len(descs), ws.ptr, cfg, C.byref(grp))
            L.call("fcn_memset_async", stamps.ptr, 0, stamps.nbytes, st)
            if cold:
                L.call("fcn_memset_async", flush.ptr, 1, flush.nbytes, st)      # push the operands out of L2 and the Infinity Cache
            L.call("fcn_stream_sync", st)
            L.check(lib.fcn_debug_conv_stamps(stamps.ptr, CAP))
            L.call("fcn_conv2d_fwd_group_f32", C.byref(grp), st)
            L.call("fcn_stream_sync", st)
            L.check(lib.fcn_debug_conv_stamps(None, 0))
            raw = np.empty((CAP, 32), np.uint64)
            L.call("fcn_memcpy_d2h_async", raw.ctypes.data, stamps.ptr, raw.nbytes, None)
            L.call("fcn_device_sync")
            raw = raw[:tiles]
            real = raw[:, [2 * k for k in ORDER]].astype(np.int64)
            cyc = raw[:, [2 * k + 1 for k in ORDER]].astype(np.int64)
            ok = real[:, 0] > 0
            t = (real[ok] - real[ok, 0].min()) / 100.0      # us
            clk = (cyc[ok, -1] - cyc[ok, 0]) / np.maximum((real[ok, -1] - real[ok, 0]) * 10.0, 1)      # GHz
            xcc = raw[ok, 30].astype(np.int64) & 0xF
            print("  cfg %d  %d workgroups (%d stamped)  event %.2f us%s  span %.2f us  shader clock %.2f GHz  wg per XCC %s" % (
                cfg, tiles, int(ok.sum()), us, " (cold)" if cold else "", t[:, -1].max(), float(np.median(clk)),
                np.bincount(xcc, minlength=8).tolist()))
            if os.environ.get("TIMELINE_PLACEMENT"):      # which CU did workgroup p run on?  (HW_ID: cu_id [11:8], sh_id [12], se_id [15:13])
                hw = raw[:, 31].astype(np.int64)
                cu = ((raw[:, 30].astype(np.int64) & 0xF) << 8) | ((hw >> 8) & 0xFF)
                n = len(cu)
                per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
                same = [int((cu[:n - d] == cu[d:]).sum()) for d in (8, 32, 256)]
                print("    placement: %d distinct CUs, workgroups per CU min %d max %d; workgroup p and p+8 / p+32 / p+256 on the same CU: %s of %s" % (
                    len(per_cu), per_cu.min(), per_cu.max(), same, [n - 8, n - 32, n - 256]))
            have = [i for i in range(len(NAMES)) if (real[ok, i] > 0).all()]      # (split-role configs: thread 0 is a multiplier wave and never
            prev_i = 0                                                              #  executes the loader's stamps; short K loops lack iter1..3)
            for i, nm in enumerate(NAMES):
                if i not in have:
                    continue
                col = t[:, i]
                prev = t[:, prev_i]
                prev_i = i
                print("    %-7s min %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f   | since entry p50 %6.2f max %6.2f   | step p50 %5.2f" % (
                    nm, col.min(), np.median(col), np.percentile(col, 90), col.max(), np.median(col - t[:, 0]), (col - t[:, 0]).max(),
                    np.median(col - prev)))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
