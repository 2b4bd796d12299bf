#!/bin/bash
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_hip_device_level.py tests/test_hip_configs.py tests/test_round3_parity.py tests/test_hip_fuzz.py tests/test_hip_golden.py -x -q -m gpu > gpurun_out/r5/tests6.log 2>&1
echo "tests rc=$?"; tail -8 gpurun_out/r5/tests6.log
timeout -k 10 300 python tools/debug/aperm_time.py 2>&1 | grep -v amdgpu.ids | tail -12
