#!/bin/bash
# SQ counters of the crossprod DMA kernel (run on the GPU box from the repo root).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_pbc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" \
           "SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH" \
           "GRBM_GUI_ACTIVE SQ_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/s$i -- python3 $ARGS > $OUT/s$i.log 2>&1 || echo "set $i failed: $set"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/s*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "crossprod_pbc_dma" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]
    print(f"{k:28s} {sum(v)/len(v):16.0f}   (n={len(v)})")
PY
