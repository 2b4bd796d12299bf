"""colMedians at BASELINE config 2 (every median is 0: decided by the counting pass) and on a 60 % dense operand of
positive values (every median is an order statistic of the stored values: the per-column radix select).  Run on the
GPU box."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsearray_amd import synth, _hip
from sparsearray_amd.device import DeviceCSC, colmedians
_hip.init()
dev = torch.device("cuda", 0)
for nrow, ncol, dens in ((1_000_000, 10_000, 0.01), (100_000, 2_000, 0.6)):
    cp, ri, v = synth.random_device_csc(nrow, ncol, dens, seed=7, device=dev)
    if dens > 0.5:
        v = v.abs()
    A = DeviceCSC(nrow, cp, ri, v)
    for _ in range(3): colmedians(A)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): m = colmedians(A)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100
    print(f"{nrow} x {ncol} @ {dens}: colMedians {ms:.3f} ms  {A.nnz / ms / 1e6:.0f} GNZ/s  nonzero medians {int((m != 0).sum())}", flush=True)
    del A, cp, ri, v
