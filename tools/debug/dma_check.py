import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from helpers import random_csc
from sparsearray_amd.device import DeviceCSC, PbcPlan, _lib

def run(nrow, ncol, K, cbw, dens=0.005, reps=3):
    cp, ri, v = random_csc(nrow, ncol, dens, seed=31)
    import scipy.sparse as sp
    M = sp.csc_matrix((v, ri, cp), shape=(nrow, ncol))
    y = np.random.default_rng(32).uniform(-1, 1, (nrow, K))
    want = (M.T @ y)
    A = DeviceCSC.from_host(nrow, cp, ri, v)
    plan = PbcPlan(A, K, cbw, 16, 7)
    Yd = torch.as_tensor(np.ascontiguousarray(y.T), device="cuda")
    out = torch.zeros((K, ncol), dtype=torch.float64, device="cuda")
    for rep in range(reps):
        out.zero_()
        plan.run(Yd, nrow, out)
        torch.cuda.synchronize()
        got = out.cpu().numpy().T
        err = np.abs(got - want)
        bad = err > 1e-9
        print(f"nrow={nrow} ncol={ncol} K={K} cbw={cbw} rep={rep}: bad {bad.sum()} of {bad.size}, max err {err.max():.3e}")
        if bad.any():
            cols = np.unique(np.nonzero(bad)[0]); ks = np.unique(np.nonzero(bad)[1])
            print("   bad cols:", cols[:20], "... n=", len(cols), " bad k:", ks[:20], "n=", len(ks))
            c = cols[0]
            print("   col", c, "got", got[c, ks[:4]], "want", want[c, ks[:4]])

for args in [(20077, 2100, 70, 32), (20077, 2100, 64, 32), (20096, 2100, 64, 32), (20077, 500, 70, 32), (20077, 2100, 70, 40), (70000, 1100, 128, 32)]:
    run(*args)
