#!/bin/bash
# SQ counters of the transposition kernels (run on the GPU box from the repo root)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_t
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/s$i -- python3 $ROOT/tools/debug/t_only.py > $OUT/s$i.log 2>&1 || echo "set $i failed: $set"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/s*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "transpose_" in k:
            agg[k.split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in agg:
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"   {c:26s} {sum(v)/len(v):16.0f}")
PY
