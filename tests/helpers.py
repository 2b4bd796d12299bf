"""Shared helpers: golden-case decoding, comparison rules, input builders."""
from __future__ import annotations

import json
import os
import warnings

import numpy as np

from sparsearray_amd import NA_integer, NA_real, SVT_SparseArray, is_NA_real

HERE = os.path.dirname(os.path.abspath(__file__))

_golden = None


def golden_cases():
    global _golden
    if _golden is None:
        with open(os.path.join(HERE, "golden", "golden.json")) as f:
            _golden = json.load(f)["cases"]
    return _golden


def dec(o, lacunar=True):
    if isinstance(o, dict) and o.get("__nd__"):
        dt = np.float64 if o["dtype"] == "f8" else np.int32
        a = np.frombuffer(bytes.fromhex(o["hex"]), dtype=dt)
        return a.reshape(o["shape"], order="F").copy(order="F")
    if isinstance(o, dict) and o.get("__svt__"):
        return SVT_SparseArray.from_dense(dec(o["dense"]), type=o["type"],
                                          lacunar=lacunar,
                                          na_background=bool(o.get("na_background", False)))
    if isinstance(o, dict) and o.get("__dgc__"):
        return (tuple(o["dim"]), dec(o["p"]), dec(o["i"]), dec(o["x"]))
    return o


# ---------------------------------------------------------------------------
# comparison rules
#   identical: testthat::expect_identical -- same type and values; doubles
#              bit-equal except that any two NaNs of the same class (NA vs NaN)
#              match.
#   equal:     testthat::expect_equal -- tolerance 1.5e-8, NA and NaN both
#              count as "missing" (all.equal.numeric).
#   gpu:       the north-star bar for double reductions computed in a
#              different order: 1e-6 relative, NaN-class must agree, and
#              NA-class must agree wherever the reference returns NA_real_
#              explicitly (callers pass strict_na=True there).
# ---------------------------------------------------------------------------
def _as_arrays(cur, exp):
    cur = np.asarray(cur)
    exp = np.asarray(exp)
    return cur, exp


def assert_identical(cur, exp, what=""):
    cur, exp = _as_arrays(cur, exp)
    assert cur.shape == exp.shape or cur.size == exp.size == 1 or \
        (cur.size == exp.size == 0), f"{what}: shape {cur.shape} != {exp.shape}"
    cur, exp = cur.reshape(-1, order="F"), exp.reshape(-1, order="F")
    if exp.dtype == np.int32:
        assert cur.dtype == np.int32, f"{what}: dtype {cur.dtype} != int32"
        assert np.array_equal(cur, exp), f"{what}: {cur} != {exp}"
        return
    assert cur.dtype == np.float64, f"{what}: dtype {cur.dtype} != float64"
    cn, en = np.isnan(cur), np.isnan(exp)
    assert np.array_equal(cn, en), f"{what}: NaN pattern {cur} vs {exp}"
    assert np.array_equal(is_NA_real(cur), is_NA_real(exp)), \
        f"{what}: NA/NaN class differs {cur} vs {exp}"
    assert np.array_equal(cur[~cn], exp[~en]), f"{what}: {cur} != {exp}"


def assert_equal(cur, exp, tol=1.5e-8, what="", strict_na=False, atol=0.0):
    """``atol``: absolute floor for sums that cancel (the order of a device
    reduction differs from the reference's; the 1e-6 bar is relative to the
    magnitude of the terms, not of a result that cancelled to ~0)."""
    cur, exp = _as_arrays(cur, exp)
    assert cur.size == exp.size, f"{what}: size {cur.size} != {exp.size}"
    cur = cur.reshape(-1, order="F").astype(np.float64)
    exp = exp.reshape(-1, order="F").astype(np.float64)
    if np.asarray(exp).dtype == np.int32:
        pass
    cn, en = np.isnan(cur), np.isnan(exp)
    assert np.array_equal(cn, en), f"{what}: missing pattern {cur} vs {exp}"
    if strict_na:
        assert np.array_equal(is_NA_real(cur), is_NA_real(exp)), \
            f"{what}: NA/NaN class differs"
    c, e = cur[~cn], exp[~en]
    inf = np.isinf(e)
    assert np.array_equal(c[inf], e[inf]), f"{what}: infinities differ"
    c, e = c[~inf], e[~inf]
    if c.size:
        err = np.abs(c - e)
        bound = tol * np.maximum(np.abs(e), np.abs(c))
        ok = (err <= bound) | (err <= atol)
        assert ok.all(), f"{what}: max rel err " \
            f"{np.max(err / np.maximum(np.abs(e), 1e-300)):.3e} > {tol}"


def run_case(session, case, lacunar=True):
    """Run one golden case through ``session``; returns (result, warnings)."""
    args = [dec(a, lacunar) for a in case["args"]]
    kwargs = dict(case["kwargs"])
    fn = getattr(session, case["fn"])
    if case["fn"] in ("rowsum", "colsum"):
        group = kwargs.pop("group")
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            res, _ug = fn(args[0], group, **kwargs)
        return res, w
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        res = fn(*args, **kwargs)
    return res, w


def check_case(session, case, lacunar=True, gpu=False):
    import pytest
    from sparsearray_amd import SparseArrayError
    what = f"case {case['id']} {case['fn']} [{case['src']}]"
    if "error" in case:
        with pytest.raises(SparseArrayError, match=case["error"]):
            run_case(session, case, lacunar)
        return
    res, w = run_case(session, case, lacunar)
    if isinstance(res, SVT_SparseArray):         # aperm(): compare as dense arrays, and the
        for lf in res.leaves:                    # leaves must be well formed
            assert lf is None or (len(lf[0]) > 0 and np.all(np.diff(lf[0]) > 0)), what
        res = res.to_dense()
    msgs = [str(x.message) for x in w]
    want = case.get("warn")
    if want is None:
        assert not msgs, f"{what}: unexpected warning {msgs}"
    elif want != "*":
        assert any(want in m for m in msgs), f"{what}: missing warning {want!r}"
    if "expected" not in case:
        return
    exp = dec(case["expected"])
    if case["cmp"] == "identical":
        if gpu and np.asarray(exp).dtype == np.float64:
            # double results reduced in a different order on the device
            assert_equal(res, exp, tol=1e-6, what=what, strict_na=True)
        else:
            assert_identical(res, exp, what)
    else:
        assert_equal(res, exp, tol=1e-6 if gpu else 1.5e-8, what=what)


# ---------------------------------------------------------------------------
# random inputs in the style of randomSparseArray(), R/randomSparseArray.R:11-38
# ---------------------------------------------------------------------------
def signif2(x):
    x = np.asarray(x, dtype=np.float64)
    out = np.zeros_like(x)
    nz = x != 0
    mag = np.floor(np.log10(np.abs(x[nz])))
    f = 10.0 ** (1 - mag)
    out[nz] = np.round(x[nz] * f) / f
    return out


def random_csc(nrow, ncol, density, seed, dtype="double"):
    """Exactly floor(nrow*ncol*density) nonzeros placed uniformly without
    replacement; values signif(N(0,1), 2)."""
    rng = np.random.default_rng(seed)
    total = nrow * ncol
    nnz = int(total * density)
    lin = rng.choice(total, size=nnz, replace=False) if total < 2 ** 31 and total <= 5e7 \
        else np.unique(rng.integers(0, total, size=int(nnz * 1.02)))[:nnz]
    lin.sort()
    col = lin // nrow
    row = (lin % nrow).astype(np.int32)
    col_ptr = np.zeros(ncol + 1, dtype=np.int64)
    np.add.at(col_ptr, col + 1, 1)
    col_ptr = np.cumsum(col_ptr)
    if dtype == "double":
        val = signif2(rng.standard_normal(len(lin)))
        val[val == 0] = 0.01
    else:
        val = rng.integers(1, 20, size=len(lin)).astype(np.int32)
    return col_ptr, row, val
