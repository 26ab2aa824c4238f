set -e -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3
mkdir -p $OUT
export FCN_TUNE_CACHE=$OUT/tune_base.json FCN_QUIET=1
python3 tools/fwd_ops.py 32 f16 > $OUT/base_fwd_ops_32_f16.txt 2>&1
python3 tools/fwd_ops.py 1 f32 > $OUT/base_fwd_ops_1_f32.txt 2>&1
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/base_f16_stats -o run -- python3 $GRAFT_REPO_ROOT/tools/fwd_resident.py 32 f16 20 > $OUT/base_f16_run.json )
f=$(find $OUT/base_f16_stats -name '*kernel_stats.csv' | head -1); cp "$f" $OUT/base_infer32_f16_kernel_stats.csv; rm -rf $OUT/base_f16_stats
python3 bench.py > $OUT/base_bench.json 2> $OUT/base_bench.err
tail -c 3000 $OUT/base_bench.json
