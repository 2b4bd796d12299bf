#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_extras
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"][:90]
    if any(k in n for k in ("rowstats", "rowpanel", "rowsum", "groupsum", "colstats", "transpose", "iota", "row_bounds", "Radix", "onesweep")):
        print(f"{n:92s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:10.1f}")
PY
