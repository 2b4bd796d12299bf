#!/bin/bash
mkdir -p gpurun_out/r5
timeout -k 10 200 python tools/debug/whole_call_vs_step.py > gpurun_out/r5/whole_call2.log 2>&1
echo "wc rc=$?"; tail -6 gpurun_out/r5/whole_call2.log
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r5/bench2_settle.json 2>/dev/null
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline --settle-ms 0 > gpurun_out/r5/bench2_nosettle.json 2>/dev/null
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r5/bench2_settle_w5.json 2>/dev/null
python - <<'PY'
import json
for n in ("settle", "nosettle", "settle_w5"):
    j = json.loads(open(f"gpurun_out/r5/bench2_{n}.json").read().strip().splitlines()[-1])
    print(n, j["ms_per_step"], j["roofline"]["kernel_ms"], j["value"])
PY
bash tools/debug/trace_py.sh tools/debug/share_steps.py 24 > gpurun_out/r5/share_trace.txt 2>&1
cat gpurun_out/r5/share_trace.txt; tail -2 gpurun_out/trace_py.log
