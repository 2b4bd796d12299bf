"""BASELINE config 5 (2e4 x 2e4 x 64 @ 0.5 %): col/row statistics timings on the device."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsearray_amd import synth
from sparsearray_amd.device import DeviceCSC, colstats, rowsums, _lib
D = (20_000, 20_000, 64)
dev = torch.device("cuda", 0)
cp, ri, v = synth.random_device_csc(D[0], D[1] * D[2], 0.005, seed=5, device=dev)
A = DeviceCSC(D[0], cp, ri, v)
nnz = A.nnz


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = torch.empty(D[1] * D[0], dtype=torch.float64, device=dev)
ws = torch.empty(_lib().svt_dev_rowstats_ws_bytes(A.nrow, A.ncol), dtype=torch.uint8, device=dev)
for name, fn, nbytes in (
    ("colSums(dims=1) -> 2e4 x 64", lambda: colstats(A, "sum"), nnz * 8 + A.ncol * 16),
    ("colSums(dims=2) -> 64", lambda: colstats(A, "sum", inner=D[1]), nnz * 8 + A.ncol * 8),
    ("colVars(dims=1)", lambda: colstats(A, "var1"), nnz * 8 + A.ncol * 16),
    ("rowSums(dims=2) -> 2e4 x 2e4", lambda: rowsums(A, inner=D[1], out=out, ws=ws), nnz * 12 + D[0] * D[1] * 8),
):
    ms = timed(fn)
    print(f"{name:32s} {ms:8.3f} ms  {nnz / ms / 1e6:7.1f} GNZ/s  {nbytes / ms / 1e6:7.0f} GB/s (algorithmic)")

# aperm(x, c(3, 1, 2)): dim 3 becomes the leaf dimension; then "row stats along dim 3" are column
# statistics of the permuted array (R/SparseArray-matrixStats.R:122-190 does exactly that on the host)
import time
for perm in ((1, 3, 2), (3, 1, 2), (2, 1, 3)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    P, pdim = A.aperm(D, perm)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    P2, _ = A.aperm(D, perm)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"aperm(x, c{perm}) -> {pdim}: first {1e3 * (t1 - t0):.1f} ms, again {1e3 * (t2 - t1):.1f} ms "
          f"({nnz / (t2 - t1) / 1e9:.1f} GNZ/s)")
    ms = timed(lambda: colstats(P, "sum"))
    print(f"  colSums(dims=1) of the permuted array        {ms:8.3f} ms  {nnz / ms / 1e6:7.1f} GNZ/s")
    del P, P2
