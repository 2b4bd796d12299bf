#!/bin/bash
mkdir -p gpurun_out/r5
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r5/tests9.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r5/tests9.log
for i in 1 2; do timeout -k 10 200 python tools/debug/share_steps.py 2>/dev/null | grep rows; done
timeout -k 10 300 python tools/debug/nonfinite_and_rowmajor_time.py 2>&1 | grep -v amdgpu.ids | tail -6
timeout -k 10 300 python tools/debug/aperm4d_time.py 2>&1 | grep -v amdgpu.ids
