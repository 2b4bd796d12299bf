#!/bin/bash
# rocprofv3 kernel trace of one python script of this repository; prints per-kernel averages.
#   bash tools/debug/prof_py.sh tools/debug/rowstats_time.py [min_calls]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/prof_py
rm -rf $D
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/$1 $3 > $R/gpurun_out/prof_py.log 2>&1 < /dev/null
f=$(find $D -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] || { echo "no kernel_stats.csv"; tail -5 $R/gpurun_out/prof_py.log; exit 1; }
python3 - "$f" "${2:-5}" <<'PY' | tee $R/gpurun_out/prof_py_summary.txt
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1]))):
    if int(r['Calls']) >= int(sys.argv[2]):
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
rm -rf $D
