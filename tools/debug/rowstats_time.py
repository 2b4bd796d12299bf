"""Row statistics at BASELINE config 2 (1e6 x 1e4 @ 1 %) and config 5 (2e4 x 2e4 x 64 @ 0.5 %, dims = 2):
per-call device time of rowSums / rowVars' second pass / rowCountNAs / rowMins.  Run on the GPU box."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsearray_amd import synth, _hip
from sparsearray_amd.device import DeviceCSC, rowsums, colstats, rowsum
_hip.init()
dev = torch.device("cuda", 0)


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


which = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("ROWSTATS_CONFIG", "25")
if "2" not in which:
    cp = None
else:
    cp, ri, v = synth.random_device_csc(1_000_000, 10_000, 0.01, seed=7, device=dev)
if cp is not None:
  A = DeviceCSC(1_000_000, cp, ri, v)
  ms = timeit(lambda: rowsums(A))
  print(f"config 2 rowSums: {ms:.3f} ms  {A.nnz / ms / 1e6:.0f} GNZ/s  {(A.nnz * 12 + 8e6) / ms / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
  grp = torch.randint(1, 1001, (A.nrow,), dtype=torch.int32, device=dev)
  rs = torch.empty(A.ncol, 1000, dtype=torch.float64, device=dev)
  ms = timeit(lambda: rowsum(A, grp, 1000, out=rs))
  print(f"config 3 rowsum(1e3 groups): {ms:.3f} ms  {A.nnz / ms / 1e6:.0f} GNZ/s  {(A.nnz * 12 + 4e6 + 8e7) / ms / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
  ms = timeit(lambda: colstats(A, "sum"))
  print(f"config 2 colSums: {ms:.3f} ms", flush=True)
  del A, cp, ri, v
if "5" not in which:
    sys.exit(0)
cp, ri, v = synth.random_device_csc(20_000, 20_000 * 64, 0.005, seed=5, device=dev)
A = DeviceCSC(20_000, cp, ri, v)
ms = timeit(lambda: rowsums(A, inner=20_000), n=5)
print(f"config 5 rowSums(dims=2): {ms:.3f} ms  {(A.nnz * 12 + 3.2e9) / ms / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
ms = timeit(lambda: rowsums(A), n=5)
print(f"config 5 rowSums(dims=1): {ms:.3f} ms  {(A.nnz * 12) / ms / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
