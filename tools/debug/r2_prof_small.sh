#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_small -- python3 $R/bench.py --nrow 125056 --steps 50 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1
f=$(find $R/gpurun_out/prof_small -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1]))):
    if int(r['Calls']) >= 50:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
