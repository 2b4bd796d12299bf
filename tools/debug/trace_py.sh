#!/bin/bash
# Timeline of the last N kernel launches of one python script of this repository (start offsets, durations,
# gaps), from a rocprofv3 kernel trace.   bash tools/debug/trace_py.sh <script> <N> "<args>"
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/trace_py
rm -rf "$D"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/$1 $3 > $R/gpurun_out/trace_py.log 2>&1 < /dev/null
f=$(find $D -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] || { echo "no kernel_trace.csv"; tail -5 $R/gpurun_out/trace_py.log; exit 1; }
python3 - "$f" "${2:-20}" <<'PY' | tee $R/gpurun_out/trace_py_summary.txt
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
rows = rows[-int(sys.argv[2]):]
t0 = int(rows[0]['Start_Timestamp']); prev_end = t0
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:10.1f} us  gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  {r['Kernel_Name'][:70]}")
    prev_end = e
PY
rm -rf "$D"
