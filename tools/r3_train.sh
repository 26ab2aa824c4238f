#!/bin/bash
# round-3 helper: weight-gradient tests, per-launch table of the training step with the tuned weight-gradient plan, the bench's train block
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 400 python -m pytest tests/test_gpu_train_kernels.py tests/test_gpu_train.py -x -q > gpurun_out/r3/train_tests.txt 2>&1 || { tail -30 gpurun_out/r3/train_tests.txt; exit 1; }
tail -3 gpurun_out/r3/train_tests.txt
FCN_QUIET=1 timeout -k 10 300 python tools/train_profile.py 8 > gpurun_out/r3/train_profile_tuned.txt 2>&1 || { tail -30 gpurun_out/r3/train_profile_tuned.txt; exit 1; }
grep -E "^==|wgrad  " gpurun_out/r3/train_profile_tuned.txt
FCN_QUIET=1 timeout -k 10 400 python bench.py --no-cpu-baseline --no-secondary --steps 50 > gpurun_out/r3/bench_train.json 2> gpurun_out/r3/bench_train.err || { tail -30 gpurun_out/r3/bench_train.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3/bench_train.json").read().strip().splitlines()[-1])
print(json.dumps(d.get("train"), indent=0)[:1500])
PY
