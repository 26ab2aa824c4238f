set -e -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3
mkdir -p $OUT
export FCN_QUIET=1
SWEEP_F16=1 SWEEP_BATCH=32 SWEEP_CFGS=${CFGS:-15,32,35,36} TIMELINE_TOP=4 timeout -k 10 600 python3 tools/conv_timeline.py ${SHAPES:-conv2_3x3 4c_3x3 3a_A} > $OUT/${TAG:-tl_f16}.txt 2>&1
cat $OUT/${TAG:-tl_f16}.txt
