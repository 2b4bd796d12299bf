"""Differential fuzzing of `x %*% y` for two sparse operands through the host-level entry point
(svt_matmul_SVT_SVT: the row-panel kernel of kernels_spmm.hip, or -- a non-finite value or an NA anywhere, or a
y that is not sparse enough -- the dense route) against the oracle (the reference's C_crossprod2_SVT_SVT on t(x)),
NA / NaN class included: random shapes, densities, both value types, poisoned entries in either operand.
    python tools/debug/fuzz_spmm.py [ncases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (before the HIP library)
import sparsearray_amd
from sparsearray_amd import NA_real, NA_integer, SVT_SparseArray
from helpers import assert_equal, random_csc
from oracle import oracle_session

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
hip = sparsearray_amd.hip_session()
orc = oracle_session()
bad = 0
for case in range(ncases):
    nrow = int(rng.choice([1, 5, 64, 129, 1000, 8191, 8193, 20000, 70001]))
    ninner = int(rng.choice([1, 3, 40, 257, 900]))
    K = int(rng.choice([1, 2, 7, 33, 130]))
    da = float(rng.choice([0.0, 0.002, 0.02, 0.2, 0.7]))
    db = float(rng.choice([0.0, 0.01, 0.04, 0.06, 0.3]))                # <= 5 % filled takes the sparse kernel
    if nrow * ninner * da > 1.5e6:                                      # (the oracle's dense route is the slow side)
        da = 1.5e6 / (nrow * ninner)
    dtype = "integer" if rng.integers(0, 3) == 0 else "double"
    cpa, ria, va = random_csc(nrow, ninner, da, seed=int(rng.integers(1 << 30)))
    cpb, rib, vb = random_csc(ninner, K, db, seed=int(rng.integers(1 << 30)))
    if dtype == "integer":
        va = np.round(va * 100).astype(np.int32); vb = np.round(vb * 100).astype(np.int32)
        va[va == 0] = 3; vb[vb == 0] = -2
    poison = int(rng.integers(0, 4))                                  # 0: clean, 1: in x, 2: in y, 3: both
    if poison & 1 and len(va):
        va = va.copy()
        va[int(rng.integers(len(va)))] = NA_integer if dtype == "integer" else rng.choice([np.inf, -np.inf, np.nan, NA_real])
    if poison & 2 and len(vb):
        vb = vb.copy()
        vb[int(rng.integers(len(vb)))] = NA_integer if dtype == "integer" else rng.choice([np.inf, np.nan, NA_real])
    x = SVT_SparseArray.from_csc((nrow, ninner), dtype, cpa, ria, va)
    y = SVT_SparseArray.from_csc((ninner, K), dtype, cpb, rib, vb)
    try:
        # one poisoned value per operand at most: the class of every cell is pinned (NA vs NaN), see helpers
        assert_equal(hip.matmul(x, y), orc.matmul(x, y), tol=1e-11, atol=1e-12,
                     strict_na=(poison != 3 and dtype == "double") or dtype == "integer", what=f"case {case}")
    except AssertionError as e:
        bad += 1
        print(f"MISMATCH case {case}: {nrow}x{ninner} %*% {ninner}x{K} da {da} db {db} {dtype} poison {poison}: {str(e)[:200]}", flush=True)
print(f"{ncases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
