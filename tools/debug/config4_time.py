"""BASELINE config 4 at full size on ONE GPU (1e7 x 5e4 @ 0.1 %, 5e8 nonzeros): crossprod with a
dense 1e7 x 128 operand and colSums.  (The 8-GPU run shards this by columns; bench.py --gpus 8.)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsearray_amd import synth
from sparsearray_amd.device import DeviceCSC, PbcPlan, colstats
nrow, ncol, K = 10_000_000, 50_000, 128
dev = torch.device("cuda", 0)
cp, ri, v = synth.random_device_csc(nrow, ncol, 0.001, seed=4, device=dev)
A = DeviceCSC(nrow, cp, ri, v)
nnz = A.nnz


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t0 = time.perf_counter(); plan = PbcPlan(A, K); torch.cuda.synchronize()
print(f"layout build {(time.perf_counter() - t0) * 1e3:.1f} ms, nnz {nnz:.3e}")
Y = synth.random_dense(nrow, K, seed=104, device=dev)
out = torch.zeros((K, ncol), dtype=torch.float64, device=dev)
ms = timed(lambda: plan.run(Y, nrow, out))
alg = nnz * 12 + nrow * K * 8 + ncol * K * 8
print(f"crossprod(A, Y 1e7 x 128)  {ms:8.3f} ms  {nnz / ms / 1e6:6.1f} GNZ/s  {alg / ms / 1e6:6.0f} GB/s (algorithmic)")
ms = timed(lambda: colstats(A, "sum"))
print(f"colSums                    {ms:8.3f} ms  {nnz / ms / 1e6:6.1f} GNZ/s  {nnz * 8 / ms / 1e6:6.0f} GB/s (algorithmic)")
