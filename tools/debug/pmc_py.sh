#!/bin/bash
# Hardware counters of one kernel of one python script of this repository, one rocprofv3 --pmc pass per set
# (run on the GPU box from the repo root).
#   bash tools/debug/pmc_py.sh tools/debug/config4_occupancy.py gatherx "0" [sets-file]
# sets-file: one counter set per line; default = the memory-path sets below.
R=${GRAFT_REPO_ROOT:-$(pwd)}
SCRIPT=$1; KERN=$2; ARGS=$3; SETS=$4
OUT=$R/gpurun_out/pmc_py
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ -n "$SETS" ]; then mapfile -t sets < $R/$SETS; else sets=(
  "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"
  "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM"
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM"
  "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
  "TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum"
  "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_REQ_sum"
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TA_FLAT_READ_WAVEFRONTS_sum"
  "TCC_HIT_sum TCC_MISS_sum FETCH_SIZE WRITE_SIZE"
); fi
i=0
for set in "${sets[@]}"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $OUT/s$i -- python3 $R/$SCRIPT $ARGS > $OUT/s$i.log 2>&1 < /dev/null || echo "set $i failed: $set"
done
python3 - "$OUT" "$KERN" <<'PY' | tee $R/gpurun_out/pmc_py_summary.txt
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/s*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]
    print(f"{k:40s} {sum(v)/len(v):18.0f}   (n={len(v)})")
PY
find $OUT -name "*.csv" -delete
