import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import random_csc
from oracle import oracle_session
from sparsearray_amd import SVT_SparseArray
from sparsearray_amd.device import DeviceCSC, PbcPlan
orc = oracle_session()
nrow, ncol, K = 4096, 700, 64
cp, ri, v = random_csc(nrow, ncol, 0.01, seed=81)
x = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, v)
A = DeviceCSC.from_host(nrow, cp, ri, v)
plan = PbcPlan(A, K)
y = np.random.default_rng(82).uniform(-1, 1, (nrow, K))
y[nrow // 2, K - 1] = np.inf
Ycm = torch.as_tensor(np.ascontiguousarray(y.T), device="cuda")
o1 = torch.zeros((K, ncol), dtype=torch.float64, device="cuda")
plan.run(Ycm, nrow, o1); torch.cuda.synchronize()
got = o1.cpu().numpy().T
want = orc.crossprod(x, y)
print("flags", plan.ws[:32].view(torch.int32).tolist())
print("col_nf[63]", plan.ws[256:256 + 64 * 12].view(torch.int32)[[63, 64 + 63, 128 + 63]].tolist())
gn, wn = np.isnan(got), np.isnan(want)
print("nan got", gn.sum(), "want", wn.sum(), "inf got", np.isinf(got).sum(), "want", np.isinf(want).sum())
bad = np.argwhere(gn != wn)
print("mismatch cells", bad[:10], len(bad))
for c, k in bad[:5]:
    rows = ri[cp[c]:cp[c + 1]]
    print(c, k, got[c, k], want[c, k], "leaf has row 2048:", 2048 in rows, "nnz", len(rows))
