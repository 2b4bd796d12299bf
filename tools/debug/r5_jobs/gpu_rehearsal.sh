#!/bin/bash
# the N > 1 control flow at full size on one GPU: ranks over gloo, all on device 0 (not a scaling measurement)
mkdir -p gpurun_out/r5
for n in 2 4; do
  timeout -k 10 500 python bench.py --gpus $n --backend gloo --same-device --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/r5/rehearsal_n$n.json 2> gpurun_out/r5/rehearsal_n$n.err
  echo "n=$n rc=$?"; tail -c 1500 gpurun_out/r5/rehearsal_n$n.json; echo; tail -3 gpurun_out/r5/rehearsal_n$n.err
done
