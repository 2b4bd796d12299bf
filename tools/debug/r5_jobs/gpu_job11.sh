#!/bin/bash
mkdir -p gpurun_out/r5
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r5/tests11.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r5/tests11.log
for i in 1 2; do timeout -k 10 200 python tools/debug/share_steps.py 2>/dev/null | grep rows; done
SHARE=1 timeout -k 10 200 python tools/debug/share_steps.py 2>/dev/null | grep rows
bash tools/debug/trace_py.sh tools/debug/share_steps.py 6 2>&1 | tail -7
