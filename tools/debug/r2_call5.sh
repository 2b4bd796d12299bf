#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_device_level.py -x -q -m gpu > gpurun_out/r2_tests5.log 2>&1 || { tail -20 gpurun_out/r2_tests5.log; exit 1; }
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu >> gpurun_out/r2_tests5.log 2>&1 || { tail -20 gpurun_out/r2_tests5.log; exit 1; }
tail -3 gpurun_out/r2_tests5.log
timeout -k 10 200 python tools/debug/build_time.py 2>&1 | tee gpurun_out/r2_build_time.log | tail -3
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_build2 -- python3 $GRAFT_REPO_ROOT/tools/debug/build_time.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_build2 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
