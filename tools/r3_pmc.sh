# per-kernel counters of one sweep shape: usage SHAPES=.. CFGS=.. PMC="A B C" TAG=.. bash tools/r3_pmc.sh
set -e -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $OUT
export FCN_QUIET=1 TMPDIR=/tmp SWEEP_F16=1 SWEEP_BATCH=32 SWEEP_CFGS=${CFGS:-35}
i=0
for set in "${PMC1:-TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum}" "${PMC2:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE}" "${PMC3:-FETCH_SIZE}" "${PMC4:-TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum}"; do
  i=$((i+1)); d=$OUT/pmc_$i; rm -rf $d
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o run -- python3 $GRAFT_REPO_ROOT/tools/conv_sweep.py ${SHAPES:-conv2_3x3} > /dev/null 2>$OUT/pmc_$i.err ) || { tail -5 $OUT/pmc_$i.err; continue; }
  f=$(find $d -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen=set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key=(k, r["Dispatch_Id"])
    if key not in seen: seen.add(key); n[k]+=1
for k in acc:
    if "conv" not in k: continue
    print(k, "calls", n[k], {c: round(v / n[k]) for c, v in acc[k].items()})
PY
  rm -rf $d
done 2>&1 | tee $OUT/${TAG:-pmc}.txt
