"""One rank's share of config 2a at N = 8 (rank 0's row block of the bench's 8-block matrix): 60 steps, for a kernel
trace (tools/debug/trace_py.sh tools/debug/share_steps.py 24) -- which launches a step holds and the gaps between them."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsearray_amd import synth, parallel as par
from sparsearray_amd.device import DeviceCSC
dev = torch.device("cuda", 0)
nrow, ncol, K = 1_000_000, 10_000, 128
nshare = int(os.environ.get("SHARE", "8"))
cp, ri, v, (r0, r1) = synth.random_device_csc_blocked(nrow, ncol, 0.01, seed=1, device=dev, nblocks=8, first=0, last=8 // nshare)
Y = synth.random_dense_blocked(nrow, K, seed=101, device=dev, nblocks=8, first=0, last=8 // nshare)
A = DeviceCSC(r1 - r0, cp, ri, v)
sc = par.ShardedCrossprod(A, K)
for _ in range(10):
    sc.step(Y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    sc.step(Y)
torch.cuda.synchronize()
print(f"rows {r1 - r0}: {(time.perf_counter() - t0) / 50 * 1e3:.4f} ms per step")
