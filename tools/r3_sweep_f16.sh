set -e -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $OUT
export FCN_QUIET=1
timeout -k 10 600 python3 -m pytest tests/test_gpu_f16.py -x -q -m gpu > $OUT/t_f16.txt 2>&1 || { tail -40 $OUT/t_f16.txt; exit 1; }
tail -3 $OUT/t_f16.txt
SWEEP_F16=1 SWEEP_BATCH=32 SWEEP_CFGS=${CFGS:-15,32,33,34,35,36,37} timeout -k 10 600 python3 tools/conv_sweep.py ${SHAPES:-conv2_3x3 conv2_red 3a_A 3a_B 3b_A 3b_B 3b_3x3 4a_A 4a_B 4c_3x3 4c_B 4e_3x3 5b_A 5b_B 5b_3x3} > $OUT/${TAG:-sweep_f16_stream}.txt 2>&1
cat $OUT/${TAG:-sweep_f16_stream}.txt
