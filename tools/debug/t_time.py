"""t(A) at config 2a: wall time per call (steady state) and a check against torch."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sparsearray_amd import synth
from sparsearray_amd.device import DeviceCSC
dev = torch.device("cuda", 0)
nrow, ncol, dens = 1_000_000, 10_000, 0.01
if len(sys.argv) > 3:
    nrow, ncol, dens = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
cp, ri, v = synth.random_device_csc(nrow, ncol, dens, seed=1, device=dev)
A = DeviceCSC(nrow, cp, ri, v)
best = 1e9
for _ in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    T = A.t(); torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) * 1e3)
# check: (row, col) pairs sorted by row then col
cols = torch.repeat_interleave(torch.arange(ncol, device=dev), (cp[1:] - cp[:-1]))
key = ri.to(torch.int64) * ncol + cols
order = torch.argsort(key, stable=True)
ok = bool(torch.equal(T.row_idx.to(torch.int64), cols[order])) and bool(torch.equal(T.val, v[order]))
print(f"t(A) {nrow}x{ncol} @ {dens}: {best:.3f} ms  nnz {A.nnz}  matches torch: {ok}")
