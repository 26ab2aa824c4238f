set -e -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $OUT
export FCN_QUIET=1
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $OUT/t_all.txt 2>&1 || { tail -40 $OUT/t_all.txt; exit 1; }
tail -3 $OUT/t_all.txt
timeout -k 10 600 python3 bench.py > $OUT/bench_a.json 2> $OUT/bench_a.err || { tail -20 $OUT/bench_a.err; exit 1; }
python3 - <<'PY'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r3/bench_a.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["single_stream"] if "single_stream" in d else None)
f=d["inference_batch32"]["f16"]; print("f16 b32:", f["frames_per_s"], f["forward_ms"], f["roofline"]["frac"], f["rel_err_vs_f32"])
print("train:", d["train"]["imgs_per_s"] if "imgs_per_s" in d["train"] else d["train"].keys())
PY
