set -e -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $OUT
export FCN_QUIET=1
timeout -k 10 900 python3 tools/fwd_ops.py 32 f16 > $OUT/${TAG:-fwd_ops_32_f16}.txt 2>&1 || { tail -30 $OUT/${TAG:-fwd_ops_32_f16}.txt; exit 1; }
cat $OUT/${TAG:-fwd_ops_32_f16}.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_f16.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -5
