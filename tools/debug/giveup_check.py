"""The fix-up launch's grid barrier gives up after ~1 s of polling and hands the product to the general kernels
(kernels_mult_pbc.hip, pbc_grid_barrier).  Tuning build: svt_dev_pbc_set_debug(7) makes every barrier's target
unreachable; the product with a dirty dense operand must still come out right (three barriers: ~3 s)."""
import os, sys, time
os.environ["SVT_HIP_TUNING"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from helpers import random_csc
from sparsearray_amd import _hip
from sparsearray_amd.device import CrossprodPlan, DeviceCSC, PbcPlan
lib = _hip.init()
dev = torch.device("cuda", 0)
nrow, ncol, K = 60_000, 1300, 64
cp, ri, v = random_csc(nrow, ncol, 0.01, seed=77)
A = DeviceCSC.from_host(nrow, cp, ri, v)
y = np.random.default_rng(78).uniform(-1, 1, (K, nrow))
y[3, int(ri[5])] = np.inf
y[9, 1234] = np.nan
Y = torch.as_tensor(y, device=dev)
want = torch.zeros((K, ncol), dtype=torch.float64, device=dev)
CrossprodPlan(A, K).run(Y, nrow, want)
plan = PbcPlan(A, K, 40, 16, 7)
for mode in (0, 7):
    lib.svt_dev_pbc_set_debug(mode)
    out = torch.full((K, ncol), 7.0, dtype=torch.float64, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    plan.run(Y, nrow, out)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    same = bool((torch.isnan(out) == torch.isnan(want)).all())
    fin = torch.isfinite(want)
    err = float((out[fin] - want[fin]).abs().max())
    print(f"debug {mode}: {dt * 1e3:.1f} ms, NaN pattern equal {same}, max abs diff of the finite cells {err:.2e}", flush=True)
lib.svt_dev_pbc_set_debug(0)
