"""Where a workgroup of `A %*% Y` (BASELINE config 2b: the product kernel on the layout of t(A),
3126 workgroups of 79 panels) spends its cycles: prologue / panel loop / epilogue of three
workgroups, printed by the tuning build (make -C sparsearray_amd/csrc TUNING=1; SVT_HIP_TUNING=1).
CBW = 32 (the instrumented kernel exists for <= 32 columns per wavefront)."""
import os, sys, time
os.environ["SVT_HIP_TUNING"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsearray_amd import synth, _hip
from sparsearray_amd.device import DeviceCSC, PbcPlan
lib = _hip.init()
dev = torch.device("cuda", 0)
nrow, ncol, K = 1_000_000, 10_000, 128
cp, ri, v = synth.random_device_csc(nrow, ncol, 0.01, seed=1, device=dev)
A = DeviceCSC(nrow, cp, ri, v)
At = A.t()
Y = torch.rand(K, ncol, dtype=torch.float64, device=dev) * 2 - 1
out = torch.empty(K, nrow, dtype=torch.float64, device=dev)
for cbw in (32, 40):
    plan = PbcPlan(At, K, cbw, 16, 7)
    for _ in range(3):
        plan.run(Y, ncol, out, 1, nrow)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        plan.run(Y, ncol, out, 1, nrow)
    torch.cuda.synchronize()
    print(f"CBW {cbw}: {(time.perf_counter() - t0) * 100:.3f} ms per product", flush=True)
    if True:
        lib.svt_dev_pbc_set_debug(3)
        plan.run(Y, ncol, out, 1, nrow)
        torch.cuda.synchronize()
        lib.svt_dev_pbc_set_debug(0)
    del plan
