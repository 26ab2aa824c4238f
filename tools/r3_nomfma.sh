set -e -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $OUT
export FCN_QUIET=1
for lib in libfcnhip.so libfcnhip_nomfma.so; do
echo "== $lib"
FCN_LIB_PATH=$GRAFT_REPO_ROOT/fcn_object_detector_amd/$lib SWEEP_F16=1 SWEEP_BATCH=32 SWEEP_CFGS=15,32,34,35,36 timeout -k 10 300 python3 tools/conv_sweep.py conv2_3x3 4c_3x3 3a_A 5b_A
done > $OUT/nomfma_f16.txt 2>&1
cat $OUT/nomfma_f16.txt
