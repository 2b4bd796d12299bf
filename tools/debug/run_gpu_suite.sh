#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/r2_tests4.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2_tests4.log
tail -12 gpurun_out/r2_tests4.log
