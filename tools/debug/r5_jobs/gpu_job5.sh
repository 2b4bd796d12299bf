#!/bin/bash
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_round5_parity.py -x -q -m gpu -s > gpurun_out/r5/tests5.log 2>&1
echo "tests rc=$?"; grep -E "spread|passed|failed|Error|error" gpurun_out/r5/tests5.log | tail -12
timeout -k 10 200 python tools/debug/median_time.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 500 python bench.py > gpurun_out/r5/bench5.json 2> gpurun_out/r5/bench5.err
echo "bench rc=$?"; python - <<'PY'
import json
j = json.loads(open("gpurun_out/r5/bench5.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["value"], j["roofline"]["kernel_ms"], j["roofline"]["frac"])
e = j["extras"]
for k in ("crossprod_whole_call", "rank_share_ms_per_step", "rank_share_speedup_before_the_collective", "first_call_from_resident_csc", "matmul_A_Y(2b)", "svt_x_svt2(3)", "host_entry_point_ms", "colMedians", "rowsum_1e3_groups"):
    print(k, e.get(k))
print(j["cpu_baseline"])
PY
