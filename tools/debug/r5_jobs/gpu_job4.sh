#!/bin/bash
mkdir -p gpurun_out/r5
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r5/tests4.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r5/tests4.log
timeout -k 10 200 python tools/debug/median_time.py 2>&1 | grep -v amdgpu.ids
for i in 1 2; do timeout -k 10 200 python tools/debug/share_steps.py 2>/dev/null | grep rows; done
SHARE=4 timeout -k 10 200 python tools/debug/share_steps.py 2>/dev/null | grep rows
SHARE=1 timeout -k 10 200 python tools/debug/share_steps.py 2>/dev/null | grep rows
bash tools/debug/trace_py.sh tools/debug/share_steps.py 12 > gpurun_out/r5/share_trace3.txt 2>&1; cat gpurun_out/r5/share_trace3.txt
