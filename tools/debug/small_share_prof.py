"""Where the fixed ~50 us of a product launch go (what a rank of an 8-GPU run sees: 125000 rows, 122 panels per
workgroup): prologue / panel loop / epilogue cycles of three workgroups, tuning build (make TUNING=1;
SVT_HIP_TUNING=1), at 1/8, 1/4 and all of the rows."""
import os, sys, time
os.environ["SVT_HIP_TUNING"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsearray_amd import synth, _hip
from sparsearray_amd.device import DeviceCSC, PbcPlan
lib = _hip.init()
dev = torch.device("cuda", 0)
ncol, K = 10_000, 128
for nrow in (125_000, 250_000, 1_000_000):
    cp, ri, v = synth.random_device_csc(nrow, ncol, 0.01, seed=1, device=dev)
    A = DeviceCSC(nrow, cp, ri, v)
    Y = torch.rand(K, nrow, dtype=torch.float64, device=dev) * 2 - 1
    out = torch.empty(K, ncol, dtype=torch.float64, device=dev)
    plan = PbcPlan(A, K, 40, 16, 7)
    for _ in range(3):
        plan.run_phase(1, Y, nrow, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        plan.run_phase(1, Y, nrow, out)
    e1.record(); torch.cuda.synchronize()
    print(f"rows {nrow}: product kernel {e0.elapsed_time(e1) / 20:.4f} ms", flush=True)
    lib.svt_dev_pbc_set_debug(3)
    plan.run_phase(1, Y, nrow, out)
    torch.cuda.synchronize()
    lib.svt_dev_pbc_set_debug(0)
    del plan, A, Y, out
