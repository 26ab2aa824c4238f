#!/bin/bash
# Round profile of `bench.py` on the GPU box: kernel-trace statistics, the two HBM counter passes and the matrix-core counter
# pass (separate runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes), summarised into gpurun_out/<tag>_*.json / .csv;
# copy the summaries into profiles/ afterwards.  usage: tools/profile_bench.sh <tag> [bench args]
set -e -o pipefail
TAG=${1:-r03}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export FCN_TUNE_CACHE=$OUT/${TAG}_tune.json
ARGS="--steps 50 --warmup 5 --no-cpu-baseline --no-secondary $*"
cd "$ROOT"
python3 bench.py $ARGS > "$OUT/${TAG}_bench_plain.json"            # fills the tune cache: profiled runs replay the plan
# ... and the plan of the one-frame-in-flight mode (no LDS cap: its own cache keys).  The tuner decides by timing and two tunings of the same
# 20 launches differ by up to 1 % in the frame (round 4: 243.1 .. 246.1 us on one box): the mode is tuned THREE times, each into a cache of its
# own that starts from the four-in-flight plan, and the plan with the shortest frame (device time of 200 forwards) is the one the profiled passes
# replay.  All three lines are kept ($OUT/${TAG}_tune1_*.json).
BASE_CACHE=$FCN_TUNE_CACHE
best=""; best_ms=""
for i in 1 2 3; do
    cp "$BASE_CACHE" "$OUT/${TAG}_tune1_$i.cache.json"
    FCN_TUNE_CACHE="$OUT/${TAG}_tune1_$i.cache.json" python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-secondary $* --in-flight 1 --no-train --no-io-region > "$OUT/${TAG}_tune1_$i.json"
    ms=$(python3 -c "import json,sys; print(json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['single_stream']['device_ms_per_step'])" "$OUT/${TAG}_tune1_$i.json")
    echo "one-frame-in-flight tuning $i: $ms ms per frame"
    if [ -z "$best" ] || python3 -c "import sys; sys.exit(0 if float(sys.argv[1]) < float(sys.argv[2]) else 1)" "$ms" "$best_ms"; then best=$i; best_ms=$ms; fi
done
echo "kept tuning $best ($best_ms ms)"
cp "$OUT/${TAG}_tune1_$best.cache.json" "$BASE_CACHE"
export TMPDIR=/tmp
# Order (round 3, after the advisor's note): the plain-launch passes FIRST - they have never failed - and every pass guarded, so that a
# failure of a later pass (the hipGraph replay under the profiler crashed once in round 1) cannot cost the data of
# the others.  PYTHONFAULTHANDLER keeps a backtrace in $OUT/<pass>.err if one dies.
prof() {      # prof <dir> <rocprofv3 options...> -- <program...>   (the program itself follows `--`: never a wrapper)
    local d=$1; shift
    rm -rf "$OUT/$d"
    local err="$OUT/$d.$(date +%H%M%S).err"      # never overwritten: a failing pass keeps its traceback (round 3 lost one to a later clean run)
    ( cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv -d "$OUT/$d" -o run "$@" ) 2> "$err" || { echo "pass $d FAILED ($err):"; tail -5 "$err"; cp "$err" "$OUT/FAILED_$(basename "$err")"; return 0; }
}
ONE="$ARGS --in-flight 1"      # ONE frame in flight: per-kernel durations only mean something when launches do not overlap
# kernel-trace passes run 4 x the frames of the counter passes: rocprofv3's per-kernel AVERAGES (what roofline_from_profile.py divides)
# carry the first frames of a process (clock ramp, cold code) - 1.8 % above the median at 50 steps, within 0.5 % at 200
TRACE="--steps 200 --warmup 5 --no-cpu-baseline --no-secondary $*"
export PYTHONFAULTHANDLER=1
export FCN_NO_GRAPH=1          # plain launches: the training step, the counter passes, and a kernel trace of the forward
prof ${TAG}_pstats --stats -- python3 "$ROOT/bench.py" $TRACE --in-flight 1 --no-train --trace-clean > "$OUT/${TAG}_bench_plain_under_rocprof.json"
prof ${TAG}_tstats --stats -- python3 "$ROOT/bench.py" $ONE --trace-clean > "$OUT/${TAG}_train_under_rocprof.json"      # whole steps only: train.roofline is derived from this trace
prof ${TAG}_fetch --pmc FETCH_SIZE -- python3 "$ROOT/bench.py" $ONE --no-train > /dev/null
prof ${TAG}_write --pmc WRITE_SIZE -- python3 "$ROOT/bench.py" $ONE --no-train > /dev/null
# matrix-core counters (SQ block, own pass): busy cycles of the MFMA pipes against the SQ's busy cycles, MFMA op counts
prof ${TAG}_mfma --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- python3 "$ROOT/bench.py" $ONE --no-train > /dev/null
prof ${TAG}_tmfma --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- python3 "$ROOT/bench.py" $ONE > /dev/null
# BASELINE configs[4]: batch-32 half-float forward - a CLEAN kernel trace first, then HBM bytes per forward and the matrix cores
N32=20
python3 tools/fwd_resident.py 32 f16 2 > /dev/null      # fills the tune cache for the batch-32 f16 plan
prof ${TAG}_f16_stats --stats -- python3 "$ROOT/tools/fwd_resident.py" 32 f16 $N32 > "$OUT/${TAG}_infer32_f16_trace_run.json"
prof ${TAG}_f16_fetch --pmc FETCH_SIZE -- python3 "$ROOT/tools/fwd_resident.py" 32 f16 $N32 > "$OUT/${TAG}_infer32_f16_run.json"
prof ${TAG}_f16_write --pmc WRITE_SIZE -- python3 "$ROOT/tools/fwd_resident.py" 32 f16 $N32 > /dev/null
prof ${TAG}_f16_mfma --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- python3 "$ROOT/tools/fwd_resident.py" 32 f16 $N32 > /dev/null
# the hipGraph replay itself, the launch path `value` is measured on: one frame in flight (what `roofline` is computed from), then four
unset FCN_NO_GRAPH
prof ${TAG}_stats --stats -- python3 "$ROOT/bench.py" $TRACE --in-flight 1 --no-train --trace-clean > "$OUT/${TAG}_bench_under_rocprof.json"
# four frames in flight: kernels only (--no-io-region).  With config 2's region in it - async copies + graph launches on four replica streams -
# hipGraphLaunch dies with SIGSEGV inside the runtime under rocprofv3 --kernel-trace, and ONLY there: three of five runs in round 3 (graph
# with memcpy nodes), again in round 4 after every kernel of the graph was launched eagerly before its capture, and again with the copies moved
# out of the graph (tracebacks: profiles/experiments/r04_graph_*segv*.txt).  One frame in flight passes under the profiler; four pass without it.
prof ${TAG}_inflight --stats -- python3 "$ROOT/bench.py" $ARGS --no-train --no-io-region > "$OUT/${TAG}_bench_inflight_under_rocprof.json"
[ -d "$OUT/${TAG}_stats" ] || { echo "graph-replay trace missing: using the plain-launch trace for the kernel statistics"; cp -r "$OUT/${TAG}_pstats" "$OUT/${TAG}_stats"; }
for pair in stats:bench_kernel_stats pstats:bench_plain_kernel_stats tstats:train_kernel_stats fetch:bench_pmc_fetch write:bench_pmc_write mfma:bench_pmc_mfma tmfma:train_pmc_mfma \
            inflight:bench_inflight_kernel_stats f16_stats:infer32_f16_kernel_stats f16_fetch:infer32_f16_pmc_fetch f16_write:infer32_f16_pmc_write f16_mfma:infer32_f16_pmc_mfma; do
    d=${pair%%:*}; o=${pair##*:}
    [ -d "$OUT/${TAG}_$d" ] && python3 tools/parse_rocprof.py "$OUT/${TAG}_$d" "$OUT/${TAG}_$o.json" || true
done
for pair in stats:bench_kernel_stats tstats:train_kernel_stats inflight:bench_inflight_kernel_stats f16_stats:infer32_f16_kernel_stats; do
    d=${pair%%:*}; o=${pair##*:}
    [ -d "$OUT/${TAG}_$d" ] || continue      # (a pass that failed leaves no directory: that must not end the script in front of the clean-up)
    f=$(find "$OUT/${TAG}_$d" -name '*kernel_stats.csv' | head -1) || true; [ -n "$f" ] && cp "$f" "$OUT/${TAG}_$o.csv"
done
python3 - "$OUT" "$TAG" $N32 <<'PY'
import json, sys
out, tag, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
try:
    f = json.load(open("%s/%s_infer32_f16_pmc_fetch.json" % (out, tag))); w = json.load(open("%s/%s_infer32_f16_pmc_write.json" % (out, tag)))
    run = json.loads(open("%s/%s_infer32_f16_run.json" % (out, tag)).read().strip().splitlines()[-1])
    total = run["forwards_total"]
    # every kernel of the run (layout converters of the one upload included: < 0.1 %), divided by the forwards it ran
    b = (f["all_kernels_hbm"]["read_bytes_total"] + w["all_kernels_hbm"]["write_bytes_total"]) / total
    json.dump({"bytes_per_forward": round(b), "read_bytes_per_forward": round(f["all_kernels_hbm"]["read_bytes_total"] / total),
               "write_bytes_per_forward": round(w["all_kernels_hbm"]["write_bytes_total"] / total), "forwards": total, "run": run,
               "kernel_source_hash": f.get("kernel_source_hash"),
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) around tools/fwd_resident.py 32 f16 %d; read = 2 x FETCH_SIZE" % n},
              open("%s/%s_infer32_f16_hbm.json" % (out, tag), "w"), indent=1)
except Exception as e:
    print("infer32 summary failed:", e)
PY
# the raw traces are large: keep the summaries only
for d in stats pstats tstats fetch write mfma tmfma inflight f16_stats f16_fetch f16_write f16_mfma; do rm -rf "$OUT/${TAG}_$d"; done
ls -la "$OUT" | grep "${TAG}_"
