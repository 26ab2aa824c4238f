#!/bin/bash
# round-3 helper: the streaming convolution without its loads / MFMAs / fragment reads / stores (elimination builds; results are wrong)
mkdir -p gpurun_out/r3
export FCN_QUIET=1 SWEEP_F16=1 SWEEP_BATCH=32
for lib in libfcnhip.so libfcnhip_s_NOLOAD.so libfcnhip_s_NOMFMA.so libfcnhip_s_NOREAD.so libfcnhip_s_NOREADMFMA.so libfcnhip_s_NOSTORE.so; do
echo "== $lib"
FCN_LIB_PATH=$GRAFT_REPO_ROOT/fcn_object_detector_amd/$lib SWEEP_CFGS=${CFGS:-34,36,40} timeout -k 10 200 python3 tools/conv_sweep.py ${SHAPES:-3a_A 4a_A 5b_A conv2_red} || exit 1
done > gpurun_out/r3/${TAG:-sexp}.txt 2>&1
cat gpurun_out/r3/${TAG:-sexp}.txt
