"""colMedians() / rowMedians(): the reference computes them in R, one leaf at a time
(R/SparseArray-matrixStats.R:690-815).  The oracle restates that R code (oracle/oracle.py);
both it and the HIP path are checked against the plain definition, median of the dense column
(with R's rule: any NA/NaN -> NA unless na.rm)."""
import numpy as np
import pytest

from helpers import assert_equal
from sparsearray_amd import NA_integer, NA_real, SVT_SparseArray, is_NA_real


def _dense_colmedians(a, na_rm):
    a = np.asarray(a, dtype=np.float64)
    out = np.empty(a.shape[1])
    for j in range(a.shape[1]):
        col = a[:, j]
        miss = np.isnan(col)
        if miss.any() and not na_rm:
            out[j] = NA_real
            continue
        col = col[~miss]
        out[j] = NA_real if col.size == 0 else np.median(col)
    return out


def _cases():
    rng = np.random.default_rng(5)
    cases = []
    # man/SparseArray-matrixStats.Rd:183-187 (the 2D example object)
    m0 = np.zeros((6, 4), dtype=np.int32)
    m0.reshape(-1, order="F")[np.array([1, 2, 8, 10, 15, 16, 17, 24]) - 1] = np.arange(1, 9) * 10
    m0 = m0.reshape((6, 4), order="F") if m0.flags.f_contiguous else np.asfortranarray(m0)
    m0 = np.asfortranarray(m0)
    m0[4, 1] = NA_integer
    cases.append(("man page m0", m0, "integer"))
    for nrow in (1, 2, 7, 8):
        a = np.round(rng.normal(size=(nrow, 40)), 1)
        a[rng.random(a.shape) < 0.5] = 0.0
        cases.append((f"small {nrow}", a, "double"))
    a = np.round(rng.normal(size=(101, 60)), 2)
    a[rng.random(a.shape) < 0.6] = 0.0
    a[:, 0] = 0.0                                   # empty leaf
    a[:, 1] = np.abs(a[:, 1]) + 1                   # all positive
    a[:, 2] = -np.abs(a[:, 2]) - 1                  # all negative
    a[:51, 3] = 5.0; a[51:, 3] = 0.0                # bare majority of positives, odd n
    a[3, 4] = np.nan
    a[5, 5] = NA_real
    a[7, 6] = np.inf
    a[:, 7] = np.nan                                # nothing left under na.rm
    a[9, 8] = -np.inf
    cases.append(("mixed 101", a, "double"))
    b = a[:100].copy()                              # even n: means of two middle values
    b[:50, 9] = 2.0; b[50:, 9] = 0.0                # exactly half positive
    b[:50, 10] = -2.0; b[50:, 10] = 0.0             # exactly half negative
    b[:50, 11] = -2.0; b[50:, 11] = 3.0             # half / half, no zeros
    cases.append(("mixed 100", b, "double"))
    c = rng.integers(-5, 6, (64, 30)).astype(np.int32)
    c[rng.random(c.shape) < 0.5] = 0
    c[2, 3] = NA_integer
    cases.append(("int 64", c, "integer"))
    return cases


def _as_float(a, type_):
    f = np.asarray(a, dtype=np.float64).copy()
    if type_ == "integer":
        f[np.asarray(a) == NA_integer] = np.nan
    return f


@pytest.mark.parametrize("na_rm", [False, True])
def test_oracle_colmedians_is_the_dense_median(oracle, na_rm):
    for name, a, type_ in _cases():
        x = SVT_SparseArray.from_dense(np.asfortranarray(a), type_)
        want = _dense_colmedians(_as_float(a, type_), na_rm)
        assert_equal(oracle.colMedians(x, na_rm=na_rm), want, tol=1e-15, strict_na=True, what=name)
        want_r = _dense_colmedians(_as_float(a, type_).T, na_rm)
        assert_equal(oracle.rowMedians(x, na_rm=na_rm), want_r, tol=1e-15, strict_na=True, what=name + " rows")
    x0 = SVT_SparseArray((0, 3), "double", [None] * 3)
    assert is_NA_real(oracle.colMedians(x0)).all()
    with pytest.raises(Exception, match="only supports 2D"):
        oracle.colMedians(SVT_SparseArray((2, 2, 2), "double", [None] * 4))


@pytest.mark.gpu
@pytest.mark.parametrize("na_rm", [False, True])
def test_hip_colmedians(hip, oracle, na_rm):
    for name, a, type_ in _cases():
        x = SVT_SparseArray.from_dense(np.asfortranarray(a), type_)
        assert_equal(hip.colMedians(x, na_rm=na_rm), oracle.colMedians(x, na_rm=na_rm),
                     tol=1e-15, strict_na=True, what=name)
        assert_equal(hip.rowMedians(x, na_rm=na_rm), oracle.rowMedians(x, na_rm=na_rm),
                     tol=1e-15, strict_na=True, what=name + " rows")
    x0 = SVT_SparseArray((0, 3), "double", [None] * 3)
    assert is_NA_real(hip.colMedians(x0)).all()
    with pytest.raises(Exception, match="only supports 2D"):
        hip.colMedians(SVT_SparseArray((2, 2, 2), "double", [None] * 4))


@pytest.mark.gpu
def test_hip_colmedians_long_columns(hip):
    rng = np.random.default_rng(6)
    a = np.round(rng.normal(size=(40_001, 120)), 3)
    a[rng.random(a.shape) < 0.7] = 0.0
    a[:, 5] = np.abs(a[:, 5]) + 0.5
    a[11, 7] = np.nan
    x = SVT_SparseArray.from_dense(np.asfortranarray(a), "double")
    for na_rm in (False, True):
        assert_equal(hip.colMedians(x, na_rm=na_rm), _dense_colmedians(a, na_rm), tol=1e-15, strict_na=True)
