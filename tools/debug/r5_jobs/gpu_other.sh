#!/bin/bash
O=gpurun_out/r5; mkdir -p $O
{
for sc in config2b_time.py config3_calls.py median_time.py aperm_time.py aperm4d_time.py; do
  echo "==== tools/debug/$sc (events, then rocprofv3 --kernel-trace --stats per-kernel averages)"
  timeout -k 10 300 python3 tools/debug/$sc 2>&1 | grep -v amdgpu.ids
  bash tools/debug/prof_py.sh tools/debug/$sc 3 2>&1 | grep -v "amdgpu.ids"
done
} > $O/r05_other_kernels.txt 2>&1
grep -c . $O/r05_other_kernels.txt
