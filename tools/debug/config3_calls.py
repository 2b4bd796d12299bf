"""BASELINE config 3 per call and prepared: svt %*% svt2 (row-panel kernel) and rowsum with 1e3 groups, HIP events.
usage: config3_calls.py [reps]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sparsearray_amd import synth
from sparsearray_amd.device import DeviceCSC, RowsumPlan, SpmmPlan, matmul_csc_csc, rowsum, _lib
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
N, M, K = 1_000_000, 10_000, 128
cp, ri, v = synth.random_device_csc(N, M, 0.01, seed=1, device=dev)
A = DeviceCSC(N, cp, ri, v)
bcp, bri, bv = synth.random_device_csc(M, K, 0.01, seed=303, device=dev)
B = DeviceCSC(M, bcp, bri, bv)
out = torch.empty((K, N), dtype=torch.float64, device=dev)
ws = torch.empty(_lib().svt_dev_matmul_csc_csc_ws_bytes(A.handle), dtype=torch.uint8, device=dev)
grp = torch.randint(1, 1001, (N,), device=dev, dtype=torch.int32)


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


alg3 = A.nnz * 12 + B.nnz * 12 + N * K * 8
ms = timed(lambda: matmul_csc_csc(A, B, out=out, ws=ws))
print(f"svt %*% svt2 per call      {ms:.3f} ms  {A.nnz / ms / 1e6:6.1f} GNZ/s  frac {alg3 / ms / 1e6 / 8000:.3f}", flush=True)
sp = SpmmPlan(A)
out2 = torch.empty_like(out)
ms = timed(lambda: sp.run(B, out=out2))
print(f"svt %*% svt2 prepared      {ms:.3f} ms  {A.nnz / ms / 1e6:6.1f} GNZ/s  frac {alg3 / ms / 1e6 / 8000:.3f}   "
      f"max |diff| {float((out - out2).abs().max()):.2e}", flush=True)
ms = timed(lambda: SpmmPlan(A))
print(f"  prepare(A)               {ms:.3f} ms", flush=True)
del out2, sp
algr = A.nnz * 12 + N * 4 + 1000 * M * 8
o1 = rowsum(A, grp, 1000)
ms = timed(lambda: rowsum(A, grp, 1000, out=o1))
print(f"rowsum 1e3 groups per call {ms:.3f} ms  {A.nnz / ms / 1e6:6.1f} GNZ/s  frac {algr / ms / 1e6 / 8000:.3f}", flush=True)
rp = RowsumPlan(A, grp, 1000)
o2 = rp.run()
ms = timed(lambda: rp.run(out=o2))
algp = A.nnz * 10 + 1000 * M * 8
print(f"rowsum prepared            {ms:.3f} ms  {A.nnz / ms / 1e6:6.1f} GNZ/s  frac {algr / ms / 1e6 / 8000:.3f} of the per-call bytes "
      f"({algp / ms / 1e6 / 8000:.3f} of its own 10 B/nz)   max |diff| {float((o1 - o2).abs().max()):.2e}", flush=True)
ms = timed(lambda: RowsumPlan(A, grp, 1000))
print(f"  prepare(A, group)        {ms:.3f} ms", flush=True)
