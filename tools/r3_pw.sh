#!/bin/bash
# round-3 helper: half-float pooling / LRN launches one by one, for the product build and the named experiment builds
mkdir -p gpurun_out/r3
export FCN_QUIET=1
for lib in ${LIBS:-libfcnhip_base.so libfcnhip_nodot2.so libfcnhip.so}; do
echo "== $lib"
FCN_LIB_PATH=$GRAFT_REPO_ROOT/fcn_object_detector_amd/$lib timeout -k 10 200 python3 tools/pw_bench.py 32 || exit 1
done > gpurun_out/r3/${TAG:-pw}.txt 2>&1
cat gpurun_out/r3/${TAG:-pw}.txt
