#!/usr/bin/env python
"""Derive bench.py's `roofline.frac` for the f32 batch-1 convolution family from the committed rocprofv3 summaries ALONE.

usage: roofline_from_profile.py <tag> [train batch]   (reads profiles/<tag>_bench_kernel_stats.csv [+ <tag>_bench_pmc_mfma.json] and
                                                       profiles/<tag>_train_kernel_stats.csv; writes profiles/<tag>_bench_roofline.json and
                                                       profiles/<tag>_train_roofline.json)

  conv_us_per_frame = sum of TotalDurationNs of the forward convolution kernels (conv_fwd_group*, conv_fwd_one*, conv_first7*,
                      conv_dot1x1*) / frames, frames = calls of the first-layer kernel (it runs once per forward)
  achieved          = 15.608 GFLOP / conv_us_per_frame          frac = achieved / 157.3 TFLOP/s
                      (when the trace holds pool3_lrn5_conv1x1_kernel - conv2/3x3_reduce folded into the pool1 + norm1 pass - that
                      kernel's time is NOT in the family and the 0.103 GFLOP of conv2/3x3_reduce are NOT in the numerator: 15.505 GFLOP)
  mfma_busy_frac    = sum of SQ_VALU_MFMA_BUSY_CYCLES / 1024 (cycles the average matrix pipe of the chip was busy, per frame)
                      / (conv_us_per_frame x clock); the clock is an ASSUMPTION written into the file (2.18 GHz: what the stamped
                      build measured inside these kernels in round 2; 2.4 GHz is what the 157.3 TF peak assumes)
rocprofv3's durations are in-sequence (each kernel in the cache state of a real forward); bench.py times its launches the same way
(Engine.time_ops_in_sequence), so the two agree; the `warm` figure bench.py prints beside it repeats one launch back to back."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FWD_GFLOP, PEAK_TF, CLOCK_GHZ = 15.608, 157.3, 2.18
CONV = ("conv_fwd_group", "conv_fwd_one", "conv_first7", "conv_dot1x1")
FOLDED = "pool3_lrn5_conv1x1_kernel"                 # pool1 -> norm1 -> conv2/3x3_reduce in one HBM-bound launch (csrc/pointwise.hip)
FOLDED_GFLOP = 2.0 * 112 * 112 * 64 * 64 / 1e9       # conv2/3x3_reduce at 448 x 448, batch 1 (deploy.prototxt:77-104)


def main():
    tag = sys.argv[1]
    stats = os.path.join(ROOT, "profiles", tag + "_bench_kernel_stats.csv")
    conv_ns, frames, rows, folded = 0, 0, [], None
    for r in csv.DictReader(open(stats)):
        name = r["Name"]
        if FOLDED in name:
            folded = {"kernel": FOLDED, "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2), "gflop_not_counted": round(FOLDED_GFLOP, 4),
                      "note": "conv2/3x3_reduce runs inside the pool1 + norm1 pass: neither this kernel's time nor that convolution's FLOPs are in the family"}
        if any(c in name for c in CONV) and "wgrad" not in name:
            conv_ns += int(r["TotalDurationNs"])
            rows.append({"kernel": name.replace("(anonymous namespace)::", "")[:90], "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2)})
            if "conv_first7" in name:
                frames += int(r["Calls"])
    if not frames:
        raise SystemExit("no first-layer kernel in %s: cannot count the frames" % stats)
    us = conv_ns / 1e3 / frames
    gflop = FWD_GFLOP - (FOLDED_GFLOP if folded else 0.0)
    out = {"source": "profiles/%s_bench_kernel_stats.csv" % tag, "frames": frames, "conv_us_per_frame": round(us, 2), "family_gflop_per_frame": round(gflop, 4),
           "achieved_tflops": round(gflop / us * 1e3, 3), "peak_tflops": PEAK_TF, "frac": round(gflop / us * 1e3 / PEAK_TF, 4),
           "kernels": rows, "clock_ghz_assumed": CLOCK_GHZ}
    if folded:
        out["folded"] = folded
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", tag + "_bench_pmc_mfma.json")))
        busy, pframes = 0.0, 0
        for k, v in pmc["counters"].items():
            if any(k.startswith(c) for c in CONV):
                busy += v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
                if k.startswith("conv_first7"):
                    pframes += v["calls"]
        if pframes:
            per_frame = busy / 1024.0 / pframes
            out["mfma_busy_cycles_per_simd_per_frame"] = round(per_frame, 1)
            out["mfma_busy_frac"] = round(per_frame / (us * CLOCK_GHZ * 1e3), 4)
            out["mfma_busy_note"] = ("SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs per frame (profiles/%s_bench_pmc_mfma.json) over conv_us_per_frame x the assumed clock; "
                                     "at 2.4 GHz the same cycles are %.4f of the time" % (tag, per_frame / (us * 2.4e3)))
            out["kernel_source_hash"] = pmc.get("kernel_source_hash")
    except (OSError, KeyError, ValueError):
        pass
    if "kernel_source_hash" not in out:
        try:
            out["kernel_source_hash"] = json.load(open(os.path.join(ROOT, "profiles", tag + "_bench_kernel_stats.json"))).get("kernel_source_hash")
        except (OSError, ValueError):
            pass
    dst = os.path.join(ROOT, "profiles", tag + "_bench_roofline.json")
    json.dump(out, open(dst, "w"), indent=1)
    print("wrote", dst, "frac", out["frac"], "mfma_busy_frac", out.get("mfma_busy_frac"))
    train_roofline(tag, int(sys.argv[2]) if len(sys.argv) > 2 else 8)


def train_roofline(tag, batch):
    """bench.py's `train.roofline` from the kernel trace of the REAL two-stream training step (profiles/<tag>_train_kernel_stats.csv, a
    `bench.py --trace-clean` pass: whole steps only): the weight-gradient family = conv_wgrad_* + reduce_partials_* kernels, steps = calls of
    the solver kernel (one per step), FLOPs of a step = 2 * Cout * K * M over the 59 convolutions = 15.608 GFLOP x batch."""
    stats = os.path.join(ROOT, "profiles", tag + "_train_kernel_stats.csv")
    if not os.path.isfile(stats):
        return
    wg_ns, red_ns, steps, rows = 0, 0, 0, []
    for r in csv.DictReader(open(stats)):
        name = r["Name"]
        if "conv_wgrad" in name or "reduce_partials" in name:
            if "reduce_partials" in name:
                red_ns += int(r["TotalDurationNs"])
            else:
                wg_ns += int(r["TotalDurationNs"])
            rows.append({"kernel": name.replace("(anonymous namespace)::", "")[:90], "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2)})
        if "sgd_kernel" in name or "adam_kernel" in name:
            steps += int(r["Calls"])
    if not steps:
        print("no solver kernel in %s: cannot count the steps" % stats)
        return
    ms = (wg_ns + red_ns) / 1e6 / steps
    gflop = FWD_GFLOP * batch
    out = {"source": "profiles/%s_train_kernel_stats.csv" % tag, "steps": steps, "batch": batch, "wgrad_ms_per_step": round(ms, 4),
           "wgrad_kernels_ms_per_step": round(wg_ns / 1e6 / steps, 4), "reduce_partials_ms_per_step": round(red_ns / 1e6 / steps, 4),
           "wgrad_gflop_per_step": round(gflop, 2), "achieved_tflops": round(gflop / ms, 2), "peak_tflops": PEAK_TF, "frac": round(gflop / ms / PEAK_TF, 4),
           "kernels": rows,
           "note": "kernel durations inside the two-stream step: the weight-gradient kernels share the chip with the data-gradient stream, so their "
                   "durations are longer than when each launch is timed alone (bench.py train.roofline.isolated_launches)"}
    try:
        out["kernel_source_hash"] = json.load(open(os.path.join(ROOT, "profiles", tag + "_train_kernel_stats.json"))).get("kernel_source_hash")
    except (OSError, ValueError):
        pass
    dst = os.path.join(ROOT, "profiles", tag + "_train_roofline.json")
    json.dump(out, open(dst, "w"), indent=1)
    print("wrote", dst, "frac", out["frac"], "ms per step", out["wgrad_ms_per_step"])


if __name__ == "__main__":
    main()
