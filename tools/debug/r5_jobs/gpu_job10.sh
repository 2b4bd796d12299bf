#!/bin/bash
mkdir -p gpurun_out/r5
timeout -k 10 500 python bench.py > gpurun_out/r5/bench10.json 2> gpurun_out/r5/bench10.err
echo "bench rc=$?"; python - <<'PY'
import json
j = json.loads(open("gpurun_out/r5/bench10.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["value"], j["roofline"]["kernel_ms"], j["roofline"]["frac"])
e = j["extras"]
for k in ("crossprod_whole_call", "rank_share_ms_per_step", "rank_share_speedup_before_the_collective", "first_call_from_resident_csc", "matmul_A_Y(2b)", "svt_x_svt2(3)", "rowsum_1e3_groups"):
    print(k, e.get(k))
PY
bash tools/debug/trace_py.sh tools/debug/share_steps.py 8 2>&1 | tail -9
