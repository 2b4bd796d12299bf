#!/bin/bash
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_hip_device_level.py tests/test_round3_parity.py -x -q -m gpu > gpurun_out/r5/tests8.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r5/tests8.log
timeout -k 10 300 python tools/debug/aperm4d_time.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/debug/config3_calls.py 20 2>&1 | grep -v amdgpu.ids
