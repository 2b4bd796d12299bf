#!/bin/bash
# round 2, first GPU call: new tests, bench N=1, gloo same-device rehearsal of N=2 strong scaling
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_round2_fixes.py tests/test_hip_configs.py -x -q -m gpu > gpurun_out/r2_tests1.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2_tests1.log
tail -15 gpurun_out/r2_tests1.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err
echo "bench rc=$?"
tail -c 1500 gpurun_out/r2_bench1.json
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 10 --warmup 2 --backend gloo --same-device --no-extras > gpurun_out/r2_bench_n2.json 2> gpurun_out/r2_bench_n2.err
echo "bench n2 rc=$?"
tail -c 1200 gpurun_out/r2_bench_n2.json
