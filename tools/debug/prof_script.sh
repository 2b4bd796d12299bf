#!/bin/bash
# rocprofv3 kernel statistics of one of the timing scripts (run on the GPU box from the repo root):
#   bash tools/debug/prof_script.sh tools/debug/config5_time.py config5
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
SCRIPT=$1
TAG=${2:-script}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/$SCRIPT > $OUT/trace.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/*/*_kernel_stats.csv")[0]
print(open("$OUT/trace.log").read())
print("rocprofv3 --kernel-trace --stats, kernels of libsvt_hip.so / hipcub (all calls incl. warm-up):")
for r in csv.DictReader(open(f)):
    n = r["Name"][:100]
    if "at::" in n or "elementwise" in n or "philox" in n or "distribution" in n.lower():
        continue
    print(f"{n:102s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f}")
PY
