#!/bin/bash
# L2 counters of the gather kernel on one rank's share of BASELINE config 4 (run on the GPU box from the repo root)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_c4
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/s$i -- python3 $ROOT/bench.py --config 4 --nrow 1250000 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/s$i.log 2>&1 || echo "set $i failed: $set"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/s*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gather" in k or "pbc" in k:
            agg[k.split("(")[0][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in agg:
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"   {c:26s} {sum(v)/len(v):18.0f}  (n={len(v)})")
PY
