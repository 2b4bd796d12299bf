#!/bin/bash
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_hip_device_level.py -x -q -m gpu > gpurun_out/r5/tests3.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r5/tests3.log
timeout -k 10 300 python tools/debug/config2b_time.py > gpurun_out/r5/config2b.log 2>&1; echo "2b rc=$?"; cat gpurun_out/r5/config2b.log | grep -v amdgpu.ids
timeout -k 10 300 python tools/debug/config3_calls.py 20 > gpurun_out/r5/config3_order1.log 2>&1; echo "c3 rc=$?"; grep -v amdgpu.ids gpurun_out/r5/config3_order1.log
SVT_HIP_TUNING=1 SVT_SPMM_ORDER=0 timeout -k 10 300 python tools/debug/config3_calls.py 20 > gpurun_out/r5/config3_order0.log 2>&1; echo "c3 order0 rc=$?"; grep -v amdgpu.ids gpurun_out/r5/config3_order0.log | head -3
for i in 1 2; do
timeout -k 10 200 python tools/debug/share_steps.py 2>/dev/null | grep rows
NT=1 timeout -k 10 200 python tools/debug/share_steps.py 2>/dev/null | grep rows
done
bash tools/debug/trace_py.sh tools/debug/share_steps.py 12 > gpurun_out/r5/share_trace2.txt 2>&1; cat gpurun_out/r5/share_trace2.txt
NT=1 bash tools/debug/trace_py.sh tools/debug/share_steps.py 12 > gpurun_out/r5/share_trace2_nt.txt 2>&1; cat gpurun_out/r5/share_trace2_nt.txt
