#!/bin/bash
# round-3 helper: weight-gradient parity tests, then the sweep (tile family vs the role-split kernel)
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 300 python -m pytest tests/test_gpu_train_kernels.py -x -q -k "wgrad" > gpurun_out/r3/wg_tests.txt 2>&1 || { tail -30 gpurun_out/r3/wg_tests.txt; exit 1; }
tail -3 gpurun_out/r3/wg_tests.txt
FCN_QUIET=1 SWEEP_CFGS=${CFGS:-4} timeout -k 10 400 python tools/wgrad_sweep.py $SHAPES > gpurun_out/r3/wg_sweep.txt 2>&1
cat gpurun_out/r3/wg_sweep.txt
