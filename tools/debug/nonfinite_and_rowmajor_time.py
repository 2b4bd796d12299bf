"""BASELINE config 2a: the product with a clean dense operand, with ONE Inf in it (per-column fix-up,
kernels_mult_pbc.hip), with a whole NaN column, and with the dense operand given
by rows (tcrossprod orientation: device transposition + the same kernel)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sparsearray_amd import synth
from sparsearray_amd.device import DeviceCSC, PbcPlan

nrow, ncol, K = 1_000_000, 10_000, 128
dev = torch.device("cuda", 0)
cp, ri, v = synth.random_device_csc(nrow, ncol, 0.01, seed=1, device=dev)
A = DeviceCSC(nrow, cp, ri, v)
plan = PbcPlan(A, K)
Y = synth.random_dense(nrow, K, seed=101, device=dev)
out = torch.zeros((K, ncol), dtype=torch.float64, device=dev)


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t_clean = timed(lambda: plan.run(Y, nrow, out))
Yp = Y.clone(); Yp[5, 123_457] = float("inf")
t_inf = timed(lambda: plan.run(Yp, nrow, out))
Yn = Y.clone(); Yn[7, :] = float("nan")
t_col = timed(lambda: plan.run(Yn, nrow, out), 3)
Yrm = Y.t().contiguous()
t_try = timed(lambda: plan.run(Yrm, K, out, tr_y=True))
print(f"crossprod(A, Y) config 2a: clean {t_clean:.3f} ms; one Inf in Y {t_inf:.3f} ms ({t_inf / t_clean:.2f}x); "
      f"a whole NaN column {t_col:.3f} ms; Y given by rows (tcrossprod) {t_try:.3f} ms "
      f"({t_try / t_clean:.2f}x)")
