"""Differential fuzzing of crossprod(x) / crossprod(x, y) of two sparse operands through the host-level entry points
with the sparse-aware kernel FORCED (svt_sparse_crossprod_set_cost(0): kernels_gram.hip on t(x); a non-finite value or
an NA anywhere falls through to the dense-buffer route by the kernel's flag) against the oracle (the reference's
C_crossprod1_SVT / C_crossprod2_SVT_SVT), NA / NaN class included: random shapes from one row / one column up, densities
from empty to 70 %, skewed rows, both value types, poisoned entries, and the cell-panel form forced by a random panel
height.      python tools/debug/fuzz_gram.py [ncases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (before the HIP library)
import sparsearray_amd
from sparsearray_amd import NA_real, NA_integer, SVT_SparseArray
from sparsearray_amd.device import set_sparse_crossprod_cost, set_sparse_crossprod_panel
from helpers import assert_equal, assert_identical, random_csc
from oracle import oracle_session

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
hip = sparsearray_amd.hip_session()
orc = oracle_session()
bad = 0
set_sparse_crossprod_cost(0.0)
try:
    for case in range(ncases):
        nrow = int(rng.choice([1, 2, 63, 64, 65, 1000, 4097, 30000]))
        nx = int(rng.choice([1, 2, 15, 16, 17, 64, 129, 500, 1300]))
        ny = int(rng.choice([1, 3, 40, 257]))
        dx = float(rng.choice([0.0, 0.003, 0.03, 0.2, 0.7]))
        dy = float(rng.choice([0.0, 0.01, 0.1, 0.5]))
        if nrow * nx * dx > 6e5:                                        # (the oracle's dense-buffer walk is the slow side)
            dx = 6e5 / (nrow * nx)
        if nrow * ny * dy > 3e5:
            dy = 3e5 / (nrow * ny)
        dtype = "integer" if rng.integers(0, 3) == 0 else "double"
        cpx, rix, vx = random_csc(nrow, nx, dx, seed=int(rng.integers(1 << 30)))
        cpy, riy, vy = random_csc(nrow, ny, dy, seed=int(rng.integers(1 << 30)))
        if rng.integers(0, 4) == 0 and nrow > 8 and nx > 1:            # one heavy row: every column of x holds it
            r0 = int(rng.integers(nrow))
            dense = np.zeros((nrow, nx)); 
            for j in range(nx):
                dense[rix[cpx[j]:cpx[j + 1]], j] = vx[cpx[j]:cpx[j + 1]]
            dense[r0, :] = rng.normal(size=nx) + 3.0
            cpx = np.zeros(nx + 1, dtype=np.int64); ri_l, v_l = [], []
            for j in range(nx):
                nz = np.nonzero(dense[:, j])[0]
                ri_l.append(nz); v_l.append(dense[nz, j]); cpx[j + 1] = cpx[j] + len(nz)
            rix = np.concatenate(ri_l).astype(np.int32); vx = np.concatenate(v_l)
        if dtype == "integer":
            vx = np.round(np.asarray(vx) * 100).astype(np.int32); vy = np.round(vy * 100).astype(np.int32)
            vx[vx == 0] = 3; vy[vy == 0] = -2
        poison = int(rng.choice([0, 0, 0, 1, 2]))                         # 0: clean, 1: in x, 2: in y
        if poison == 1 and len(vx):
            vx = vx.copy()
            vx[int(rng.integers(len(vx)))] = NA_integer if dtype == "integer" else rng.choice([np.inf, -np.inf, np.nan, NA_real])
        if poison == 2 and len(vy):
            vy = vy.copy()
            vy[int(rng.integers(len(vy)))] = NA_integer if dtype == "integer" else rng.choice([np.inf, np.nan, NA_real])
        panel = None
        if rng.integers(0, 3) == 0:
            panel = (int(rng.choice([0, 1, 40])), int(rng.choice([4, 6, 9])))
            set_sparse_crossprod_panel(*panel)
        x = SVT_SparseArray.from_csc((nrow, nx), dtype, cpx, rix, vx)
        y = SVT_SparseArray.from_csc((nrow, ny), dtype, cpy, riy, vy)
        try:
            exact = dtype == "integer" and poison == 0
            for got, want, what in ((hip.crossprod(x), orc.crossprod(x), "crossprod(x)"),
                                    (hip.crossprod(x, y), orc.crossprod(x, y), "crossprod(x, y)"),
                                    (hip.crossprod(y, x), orc.crossprod(y, x), "crossprod(y, x)")):
                if exact:
                    assert_identical(got, want, what=f"case {case} {what}")
                else:
                    assert_equal(got, want, tol=1e-11, atol=1e-12, strict_na=True, what=f"case {case} {what}")
                if what == "crossprod(x)":
                    g = np.asarray(got)
                    assert np.array_equal(g, g.T, equal_nan=True), f"case {case}: crossprod(x) not symmetric"
        except AssertionError as e:
            bad += 1
            print(f"MISMATCH case {case}: {nrow}x{nx} @ {dx:.3g}, {nrow}x{ny} @ {dy:.3g} {dtype} poison {poison} panel {panel}: {str(e)[:200]}", flush=True)
        finally:
            if panel is not None:
                set_sparse_crossprod_panel(-1, -1)
finally:
    set_sparse_crossprod_cost(1.0)
print(f"{ncases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
