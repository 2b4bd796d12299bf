#!/bin/bash
# what one rank of an N-GPU strong-scaling run computes per step (1/8, 1/4, 1/2 of the rows), on one GPU, no collective;
# then the two-rank rehearsal on one device (gloo)
for nrow in 125056 250112 500224 1000000; do
  timeout -k 10 200 python bench.py --nrow $nrow --steps 50 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('rows $nrow: ms/step %.4f kernel %.4f' % (j['ms_per_step'], j['roofline']['kernel_ms']))"
done
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 --steps 10 --warmup 2 --backend gloo --same-device --no-extras > gpurun_out/r2_n2.json 2> gpurun_out/r2_n2.err
echo "n2 rc=$?"; tail -c 600 gpurun_out/r2_n2.json; tail -5 gpurun_out/r2_n2.err
