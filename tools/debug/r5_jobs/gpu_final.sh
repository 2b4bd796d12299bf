#!/bin/bash
# final check of the round: the whole GPU suite, smoke, the bench line
mkdir -p gpurun_out/r5
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r5/tests_final.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r5/tests_final.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r5/bench_final.json 2> gpurun_out/r5/bench_final.err
echo "bench rc=$?"; python - <<'PY'
import json
j = json.loads(open("gpurun_out/r5/bench_final.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["value"], j["roofline"]["kernel_ms"], j["roofline"]["frac"], j["roofline"]["fp64_frac"])
e = j["extras"]
for k in e:
    print(k, e[k])
print(j["cpu_baseline"])
PY
