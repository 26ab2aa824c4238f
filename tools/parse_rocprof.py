#!/usr/bin/env python
"""Summarise rocprofv3 CSV output (kernel trace and/or PMC counter collection) into a small JSON for profiles/.

usage: parse_rocprof.py <rocprof output dir> <out.json> [--steps N]
  * kernel trace  -> per-kernel-name calls, total / average duration
  * counter CSV   -> per-kernel-name sum of every collected counter
HBM traffic convention (MI355X_MICROARCH.md, HBM / rocprofv3): on gfx950 FETCH_SIZE counts 64 B per 128-B request for
wide coalesced reads, so read bytes = 2 * FETCH_SIZE * 1024 (FETCH_SIZE is in KiB); WRITE_SIZE * 1024 is exact.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void )?([\w:]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def derive_mfma(res):
    """Matrix-core figures where the MFMA counters were collected (rocprofv3's derived MfmaUtil has no gfx950 section).
    SQ_VALU_MFMA_BUSY_CYCLES is exact: the sum over all SIMDs of 64 cycles per v_mfma_f32_32x32x2_f32 (conv_first7_kernel:
    196 workgroups x 4 waves x 308 MFMAs x 64 = 15 454 208 per launch, the value the counter reports), so
    mfma_busy_cycles_per_simd = busy / 1024 is the time the average matrix pipe of the chip (256 CUs x 4) was busy, in shader
    cycles; divide by the kernel's duration from the kernel-trace pass x the shader clock for the busy fraction.
    GRBM_GUI_ACTIVE (sum over the 8 XCDs) is NOT that duration under --pmc: it includes ~10 us of counter set-up per
    dispatch (64 k cycles for an 18 us kernel), so `mfma_util` = busy / (GRBM_GUI_ACTIVE / 8 x 1024) is a lower bound that only
    means something for long kernels.  flops executed = MOPS x 512 (padding included)."""
    for k, v in res["counters"].items():
        calls = max(v.get("calls", 1), 1)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            v["mfma_busy_cycles_per_simd_per_call"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / calls, 1)
            if v.get("GRBM_GUI_ACTIVE"):
                v["mfma_util"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
        for c in ("SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VALU_MFMA_MOPS_F16"):
            if c in v:
                v["mfma_flops_executed_per_call"] = round(v[c] * 512.0 / calls)


def main():
    if sys.argv[1] == "--rederive":      # refresh the derived fields of an existing summary (the raw traces are not kept)
        for path in sys.argv[2:]:
            res = json.load(open(path))
            derive_mfma(res)
            json.dump(res, open(path, "w"), indent=1, sort_keys=True)
        return
    src, out = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else None
    res = {"source": os.path.basename(os.path.normpath(src)), "kernels": {}, "counters": {}}
    for path in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
        agg = collections.defaultdict(lambda: [0, 0])
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        total = sum(v[1] for v in agg.values())
        for k, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            res["kernels"][k] = {"calls": n, "total_us": round(ns / 1e3, 1), "avg_us": round(ns / n / 1e3, 2),
                                 "pct": round(100.0 * ns / max(total, 1), 2)}
    for path in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        calls = collections.defaultdict(set)
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add(r["Dispatch_Id"])
        for k, d in agg.items():
            e = res["counters"].setdefault(k, {"calls": len(calls[k])})
            e.update({c: v for c, v in d.items()})
    if steps:
        res["steps_profiled"] = steps
    try:      # which kernel sources this profile was measured on (bench.py reports `stale` when they have changed since)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import kernel_source_hash
        res["kernel_source_hash"] = kernel_source_hash()
    except Exception:
        pass
    if res["counters"]:
        calls = sum(v["calls"] for v in res["counters"].values())
        fetch_kib = sum(v.get("FETCH_SIZE", 0.0) for v in res["counters"].values())
        write_kib = sum(v.get("WRITE_SIZE", 0.0) for v in res["counters"].values())
        res["all_kernels_hbm"] = {"launches": calls, "read_bytes_total": 2.0 * fetch_kib * 1024, "write_bytes_total": write_kib * 1024,
                                  "read_bytes_per_launch": 2.0 * fetch_kib * 1024 / max(calls, 1), "write_bytes_per_launch": write_kib * 1024 / max(calls, 1),
                                  "note": "every kernel of the run; read = 2 x FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB"}
        derive_mfma(res)
    conv = {k: v for k, v in res["counters"].items() if k.startswith("conv_fwd") or k.startswith("conv_first7")}      # the forward convolution family
    if conv:
        calls = sum(v["calls"] for v in conv.values())
        fetch_kib = sum(v.get("FETCH_SIZE", 0.0) for v in conv.values())
        write_kib = sum(v.get("WRITE_SIZE", 0.0) for v in conv.values())
        res["conv_fwd_hbm"] = {"launches": calls, "read_bytes_per_launch": 2.0 * fetch_kib * 1024 / max(calls, 1),
                               "write_bytes_per_launch": write_kib * 1024 / max(calls, 1),
                               "note": "read = 2 x FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB"}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out)


if __name__ == "__main__":
    main()
