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
# every dense column in the class the leaf kernel WALKS for (more than 256 non-finite entries but
# fewer than the longest leaf holds: pbc_dirty_leaf_kernel re-walks each leaf once per such column) -- the worst
# case of the one-launch fix-up, against the ~35 ms the general kernels of rounds 2-4 took for a whole product
g = torch.Generator(device=dev); g.manual_seed(9)
for ncols_dirty in (1, 16, 128):
    Yw = Y.clone()
    for k in range(ncols_dirty):
        rows = torch.randint(0, nrow, (400,), generator=g, device=dev)
        Yw[k, rows] = float("nan")
    t_w = timed(lambda: plan.run(Yw, nrow, out), 3)
    print(f"  {ncols_dirty:3d} dense column(s) with ~400 NaN each (walked per leaf): {t_w:.3f} ms ({t_w / t_clean:.2f}x the clean product)")
    del Yw
