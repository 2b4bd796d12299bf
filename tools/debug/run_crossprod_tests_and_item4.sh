#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_device_level.py tests/test_hip_vs_oracle.py tests/test_hip_full_size.py tests/test_hip_fuzz.py -x -q -m gpu > gpurun_out/r2_tests3.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2_tests3.log
tail -12 gpurun_out/r2_tests3.log
timeout -k 10 300 python tools/debug/nonfinite_and_rowmajor_time.py > gpurun_out/r2_item4.log 2>&1
tail -3 gpurun_out/r2_item4.log
