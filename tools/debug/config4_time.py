"""BASELINE config 4 on ONE GPU (1e7 x 5e4 @ 0.1 %, Y 1e7 x 128): crossprod with the LDS-DMA layout
(40, 16, 7) and with the gather layout (40, 4, 10) that svt_dev_pbc_build(A, 0, 0, 0) picks at this
density, and colSums."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sparsearray_amd import synth
from sparsearray_amd.device import DeviceCSC, PbcPlan, colstats

nrow, ncol, K = 10_000_000, 50_000, 128
dev = torch.device("cuda", 0)
cp, ri, v = synth.random_device_csc(nrow, ncol, 0.001, seed=4, device=dev)
A = DeviceCSC(nrow, cp, ri, v)
Y = synth.random_dense(nrow, K, seed=104, device=dev)
out = torch.zeros((K, ncol), dtype=torch.float64, device=dev)
alg = A.nnz * 12 + nrow * K * 8 + ncol * K * 8


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ref = None
only = sys.argv[1] if len(sys.argv) > 1 else ""
for name, cfg in (("gather (40,4,10)", (40, 4, 10)), ("LDS-DMA (40,16,7)", (40, 16, 7))):
    if only and not name.startswith(only):
        continue
    torch.cuda.synchronize(); t0 = time.perf_counter()
    plan = PbcPlan(A, K, *cfg)
    torch.cuda.synchronize(); tb = (time.perf_counter() - t0) * 1e3
    ms = timed(lambda: plan.run(Y, nrow, out))
    print(f"crossprod {name}: layout {tb:.1f} ms, product {ms:.2f} ms, {A.nnz / ms / 1e6:.1f} GNZ/s, "
          f"{alg / ms / 1e6:.0f} GB/s = {alg / ms / 1e6 / 8000 * 100:.1f} % of 8 TB/s", flush=True)
    if ref is None:
        ref = out.clone()
    else:
        print("max |gather - dma| / max|.| =", float((out - ref).abs().max() / ref.abs().max()))
    del plan
ms = timed(lambda: colstats(A, "sum"), 5)
print(f"colSums: {ms:.3f} ms, {A.nnz * 8 / ms / 1e6:.0f} GB/s")
