#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_device_level.py -x -q -m gpu -k "gather or auto_layout" > gpurun_out/r2_tests6.log 2>&1 || { tail -30 gpurun_out/r2_tests6.log; exit 1; }
tail -3 gpurun_out/r2_tests6.log
timeout -k 10 600 python tools/debug/config4_time.py 2>&1 | tee gpurun_out/r2_config4.log | tail -6
