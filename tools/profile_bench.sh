#!/bin/bash
# Round profile of `bench.py` on the GPU box: kernel-trace statistics and the two HBM counter passes (separate runs, as
# MI355X_MICROARCH.md prescribes), summarised into gpurun_out/*.json.  usage: tools/profile_bench.sh <tag> [bench args]
set -e -o pipefail
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export FCN_TUNE_CACHE=$OUT/${TAG}_tune.json
ARGS="--steps 50 --warmup 5 --no-cpu-baseline --no-secondary $*"
cd "$ROOT"
python3 bench.py $ARGS > "$OUT/${TAG}_bench_plain.json"            # fills the tune cache: profiled runs replay the plan
# the profiled runs keep ONE frame in flight: per-kernel durations (what `roofline` is computed from) are only meaningful
# when launches do not overlap
ARGS="$ARGS --in-flight 1"
export TMPDIR=/tmp
# rocprofv3 (ROCm 7.2) segfaults inside hipGraphLaunch when the plan is replayed from the tune cache; the profiled runs
# therefore issue the same kernels as ordinary launches (identical kernels, arguments and order)
export FCN_NO_GRAPH=1
rm -rf "$OUT/${TAG}_stats" "$OUT/${TAG}_tstats" "$OUT/${TAG}_fetch" "$OUT/${TAG}_write"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats" -o run -- python3 "$ROOT/bench.py" $ARGS --no-train > "$OUT/${TAG}_bench_under_rocprof.json" )
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_tstats" -o run -- python3 "$ROOT/bench.py" $ARGS > "$OUT/${TAG}_train_under_rocprof.json" )
( cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_fetch" -o run -- python3 "$ROOT/bench.py" $ARGS --no-train > /dev/null )
( cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_write" -o run -- python3 "$ROOT/bench.py" $ARGS --no-train > /dev/null )
python3 tools/parse_rocprof.py "$OUT/${TAG}_stats" "$OUT/${TAG}_bench_kernel_stats.json"
python3 tools/parse_rocprof.py "$OUT/${TAG}_tstats" "$OUT/${TAG}_train_kernel_stats.json"
cp "$(find "$OUT/${TAG}_tstats" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_train_kernel_stats.csv"
python3 tools/parse_rocprof.py "$OUT/${TAG}_fetch" "$OUT/${TAG}_bench_pmc_fetch.json"
python3 tools/parse_rocprof.py "$OUT/${TAG}_write" "$OUT/${TAG}_bench_pmc_write.json"
cp "$(find "$OUT/${TAG}_stats" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_bench_kernel_stats.csv"
# the raw traces are large: keep the summaries only
rm -rf "$OUT/${TAG}_stats" "$OUT/${TAG}_tstats" "$OUT/${TAG}_fetch" "$OUT/${TAG}_write"
ls -la "$OUT" | grep "${TAG}_"
