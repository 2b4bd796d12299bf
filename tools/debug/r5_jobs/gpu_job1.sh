#!/bin/bash
# round 5, first GPU call: new tests, the bench line with the new extras, the whole-call discrepancy
mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest tests/test_hip_device_level.py tests/test_round3_parity.py tests/test_hip_configs.py -x -q -m gpu > gpurun_out/r5/tests1.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r5/tests1.log
timeout -k 10 400 python bench.py > gpurun_out/r5/bench1.json 2> gpurun_out/r5/bench1.err
echo "bench rc=$?"; tail -c 3000 gpurun_out/r5/bench1.json; tail -3 gpurun_out/r5/bench1.err
timeout -k 10 200 python tools/debug/whole_call_vs_step.py > gpurun_out/r5/whole_call.log 2>&1
echo "wc rc=$?"; cat gpurun_out/r5/whole_call.log
