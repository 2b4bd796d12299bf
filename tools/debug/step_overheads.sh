#!/bin/bash
# step time vs dominant-kernel time of the bench at full size and at the 1/8 row block one rank of an
# 8-GPU strong-scaling run holds, then the non-finite / row-major variants of the whole call
for nrow in 125056 1000000; do
  timeout -k 10 200 python bench.py --nrow $nrow --steps 50 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('rows $nrow: ms/step %.4f kernel %.4f' % (j['ms_per_step'], j['roofline']['kernel_ms']))"
done
timeout -k 10 300 python tools/debug/nonfinite_and_rowmajor_time.py 2>&1 | tail -1
