#!/bin/bash
# What ONE rank of the 8-GPU run of BASELINE config 4 computes, on one GPU:
# 1.25e6 rows x 5e4 columns @ 0.1 %, Y 1.25e6 x 128 -- crossprod + colSums per step, roofline in the line.
# Both layouts: the one svt_dev_pbc_build(A, 0, 0, 0) picks for this shape (5 nonzeros per 40 x 128 tile: the
# gather kernel) and the LDS-DMA layout forced.
cd ${GRAFT_REPO_ROOT:-.}
for L in "" "--cbw 40 --wpb 16 --logr 7"; do
  timeout -k 10 300 python bench.py --config 4 --nrow 1250000 --steps 10 --warmup 2 --no-cpu-baseline --no-extras $L \
    | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('layout [%s]: ms/step %.3f  kernel %s %.3f ms  GNZ/s %.1f  frac %.4f  alg bytes %.3e' % ('$L' or 'auto', j['ms_per_step'], r['kernel'], r['kernel_ms'], j['value'], r['frac'], r['algorithmic_bytes_per_launch']))"
done
