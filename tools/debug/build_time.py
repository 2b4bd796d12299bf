"""Once-per-operand costs at BASELINE config 2: panel-blocked layout build of A and of t(A), device
transposition t(A); wall time of repeated calls (the first call of each carries code-object load and
cold allocations)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sparsearray_amd import synth
from sparsearray_amd.device import DeviceCSC, PbcPlan

nrow, ncol, K = 1_000_000, 10_000, 128
dev = torch.device("cuda", 0)
cp, ri, v = synth.random_device_csc(nrow, ncol, 0.01, seed=1, device=dev)
A = DeviceCSC(nrow, cp, ri, v)


def wall(fn, reps=4):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) * 1e3)
        del r
    return out


print("layout build of A      (ms):", ["%.2f" % t for t in wall(lambda: PbcPlan(A, K))])
print("t(A)                   (ms):", ["%.2f" % t for t in wall(lambda: A.t())])
T = A.t()
print("layout build of t(A)   (ms):", ["%.2f" % t for t in wall(lambda: PbcPlan(T, K))])
