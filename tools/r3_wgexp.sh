#!/bin/bash
# round-3 helper: the role-split weight-gradient kernel without its loads / reads / MFMAs (elimination builds)
mkdir -p gpurun_out/r3
export FCN_QUIET=1
for lib in ${LIBS:-libfcnhip.so libfcnhip_ws_noload.so libfcnhip_ws_nomfma.so libfcnhip_ws_noread.so libfcnhip_ws_noread_nomfma.so}; do
echo "== $lib"
FCN_LIB_PATH=$GRAFT_REPO_ROOT/fcn_object_detector_amd/$lib SWEEP_NO_BIAS=${NOBIAS:-} SWEEP_CFGS=${CFGS:-4} timeout -k 10 200 python3 tools/wgrad_sweep.py ${SHAPES:-conv2_3x3 5b_3x3 5b_1x1 4a_1x1} || exit 1
done > gpurun_out/r3/${TAG:-wgexp}.txt 2>&1
cat gpurun_out/r3/${TAG:-wgexp}.txt
