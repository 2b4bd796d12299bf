#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_hip_device_level.py tests/test_hip_golden_fast_path.py tests/test_hip_full_size.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 400 python tools/debug/fuzz_pbc.py 150 9001 2>&1 | tail -1
for i in 1 2; do timeout -k 10 200 python tools/debug/share_steps.py 2>/dev/null | grep rows; done
bash tools/debug/trace_py.sh tools/debug/share_steps.py 4 2>&1 | tail -5
