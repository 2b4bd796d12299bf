"""Round-2 regression tests: marshalling of leaves longer than a staging buffer, eviction from the
resident set while a derived layout is being attached, and the per-addition integer overflow rule of
rowsum()/colsum() (src/rowsum_methods.c:66-84, 166-199)."""
import warnings

import numpy as np
import pytest

from helpers import assert_equal, assert_identical, random_csc
from sparsearray_amd import NA_integer, SVT_SparseArray

INT_MAX = 2147483647


def _svt(nrow, ncol, density, seed, dtype="double"):
    cp, ri, v = random_csc(nrow, ncol, density, seed, dtype)
    return SVT_SparseArray.from_csc((nrow, ncol), "double" if dtype == "double" else "integer", cp, ri, v)


def _quiet(f, *a, **k):
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        r = f(*a, **k)
    return r, [str(x.message) for x in w]


# ---- integer overflow: oracle against hand-computed values (CPU), device against oracle (GPU) ----
def _overflow_cases():
    """(dim, leaves, group, what): columns whose running sums leave int32 and come back."""
    big = INT_MAX
    cases = []
    # one column, one group: INT_MAX, +1, -1 -> the second addition overflows: NA + warning
    cases.append(((3, 1), [(np.array([0, 1, 2], np.int32), np.array([big, 1, -1], np.int32))], [1, 1, 1], "excursion"))
    # same values, the negative one first: never out of range -> INT_MAX, no warning
    cases.append(((3, 1), [(np.array([0, 1, 2], np.int32), np.array([big, -1, 1], np.int32))], [1, 1, 1], "no excursion"))
    # an NA before the overflow: NA without warning; after it: NA with warning
    cases.append(((4, 1), [(np.array([0, 1, 2, 3], np.int32), np.array([big, NA_integer, 5, 7], np.int32))],
                  [1, 1, 1, 1], "NA first"))
    cases.append(((4, 1), [(np.array([0, 1, 2, 3], np.int32), np.array([big, 5, NA_integer, 7], np.int32))],
                  [1, 1, 1, 1], "overflow first"))
    # two groups interleaved, only one of them overflows; second column harmless; negative side
    cases.append(((6, 2), [(np.array([0, 1, 2, 3, 4, 5], np.int32), np.array([-big, 3, -2, 4, 1, -9], np.int32)),
                           (np.array([1, 4], np.int32), np.array([7, 8], np.int32))],
                  [1, 2, 1, 2, 1, 2], "two groups"))
    return cases


@pytest.mark.parametrize("na_rm", [False, True])
def test_oracle_int_rowsum_overflow_known_answers(oracle, na_rm):
    """Pins the oracle's restatement of safe_int_add on hand-computed cases."""
    want = {
        ("excursion", False): ([NA_integer], True), ("excursion", True): ([NA_integer], True),
        ("no excursion", False): ([INT_MAX], False), ("no excursion", True): ([INT_MAX], False),
        ("NA first", False): ([NA_integer], False), ("NA first", True): ([NA_integer], True),   # na.rm: big + 5 overflows
        ("overflow first", False): ([NA_integer], True), ("overflow first", True): ([NA_integer], True),
    }
    for dim, leaves, group, what in _overflow_cases():
        if (what, na_rm) not in want:
            continue
        x = SVT_SparseArray(dim, "integer", leaves)
        (ans, _), msgs = _quiet(oracle.rowsum, x, group, na_rm=na_rm)
        exp, warn = want[(what, na_rm)]
        assert list(np.asarray(ans).ravel()) == exp, (what, na_rm)
        assert any("integer overflow" in m for m in msgs) == warn, (what, na_rm, msgs)


@pytest.mark.gpu
@pytest.mark.parametrize("na_rm", [False, True])
def test_int_rowsum_colsum_overflow_order(hip, oracle, na_rm):
    """The device reproduces the reference's one-addition-at-a-time rule, warning included."""
    for dim, leaves, group, what in _overflow_cases():
        x = SVT_SparseArray(dim, "integer", leaves)
        (g, _), gm = _quiet(hip.rowsum, x, group, na_rm=na_rm)
        (w, _), wm = _quiet(oracle.rowsum, x, group, na_rm=na_rm)
        assert_identical(g, w, what=f"rowsum {what}")
        assert any("overflow" in m for m in gm) == any("overflow" in m for m in wm), (what, gm, wm)
        # colsum of t(x): the same additions, cell by cell, in column order
        xt = x.t()
        (g, _), gm = _quiet(hip.colsum, xt, group, na_rm=na_rm)
        (w, _), wm = _quiet(oracle.colsum, xt, group, na_rm=na_rm)
        assert_identical(g, w, what=f"colsum {what}")
        assert any("overflow" in m for m in gm) == any("overflow" in m for m in wm), (what, gm, wm)


@pytest.mark.gpu
def test_int_rowsum_overflow_random(hip, oracle):
    """Values near the int32 edge with both signs in a few columns (those are redone in order),
    ordinary counts in the rest (parallel pass only)."""
    rng = np.random.default_rng(77)
    nrow, ncol, ng = 4000, 60, 7
    cp, ri, v = random_csc(nrow, ncol, 0.05, 78, "int")
    v = v.copy()
    for j in (3, 17, 41):
        k = slice(cp[j], cp[j + 1])
        v[k] = rng.choice(np.array([INT_MAX - 5, -(INT_MAX - 5), 11, -7, 1 << 30, -(1 << 30)], np.int32), cp[j + 1] - cp[j])
    v[cp[17] + 2] = NA_integer
    x = SVT_SparseArray.from_csc((nrow, ncol), "integer", cp, ri, v)
    group = list(rng.integers(1, ng + 1, nrow))
    for na_rm in (False, True):
        (g, _), gm = _quiet(hip.rowsum, x, group, na_rm=na_rm)
        (w, _), wm = _quiet(oracle.rowsum, x, group, na_rm=na_rm)
        assert_identical(g, w, what="rowsum")
        assert any("overflow" in m for m in gm) == any("overflow" in m for m in wm)
    gcol = list(rng.integers(1, ng + 1, ncol))
    (g, _), gm = _quiet(hip.colsum, x, gcol)
    (w, _), wm = _quiet(oracle.colsum, x, gcol)
    assert_identical(g, w, what="colsum")
    assert any("overflow" in m for m in gm) == any("overflow" in m for m in wm)


# ---- marshalling ----------------------------------------------------------------------------------
@pytest.mark.gpu
def test_upload_leaf_longer_than_a_staging_buffer(hip):
    """svt_upload() cuts its trips in nonzeros, not in leaves: a 1-D array with 5e6 nonzeros is one
    leaf of 60 MB against a 48 MiB pinned buffer (advisor finding, round 1)."""
    n, nz = 6_000_000, 5_000_000
    rng = np.random.default_rng(5)
    offs = np.sort(rng.choice(n, nz, replace=False)).astype(np.int32)
    vals = rng.integers(-50, 51, nz).astype(np.float64)
    vals[vals == 0] = 1.0
    x = SVT_SparseArray((n,), "double", [(offs, vals)])
    assert float(hip.sum(x)) == float(vals.sum())            # integers: exact in any order
    xi = SVT_SparseArray((n,), "integer", [(offs, vals.astype(np.int32))])
    assert float(hip.sum(xi)) == float(vals.sum())
    # a matrix whose middle leaf alone exceeds the buffer, short leaves either side
    small = (np.array([1, 5], np.int32), np.array([2.0, 3.0]))
    m = SVT_SparseArray((n, 3), "double", [small, (offs, vals), small])
    assert list(np.asarray(hip.colSums(m))) == [5.0, float(vals.sum()), 5.0]


@pytest.mark.gpu
def test_resident_eviction_while_attaching_layouts(hip, oracle):
    """A limit that forces an eviction exactly when t(x) / the panel-blocked layout of another
    operand is attached: the derived layout must land on its own entry (advisor finding, round 1)."""
    a = _svt(30000, 300, 0.02, 41)          # ~180k nonzeros: ~2.2 MB as CSC
    b = _svt(30000, 310, 0.02, 42)
    z = np.random.default_rng(43).uniform(-1, 1, (300, 4))
    zb = np.random.default_rng(44).uniform(-1, 1, (310, 4))
    want_a, want_b = oracle.matmul(a, z), oracle.matmul(b, zb)
    try:
        hip.resident_set_limit(5_500_000)       # holds a and b, not a, b and t(b)
        assert_equal(hip.colSums(a), oracle.colSums(a), tol=1e-9, atol=1e-11)      # a resident (older)
        assert_equal(hip.colSums(b), oracle.colSums(b), tol=1e-9, atol=1e-11)      # b resident
        assert hip.resident_stats()["entries"] == 2
        got_b = hip.matmul(b, zb)               # attaches t(b): evicts a (index 0) meanwhile
        assert_equal(got_b, want_b, tol=1e-9, atol=1e-11)
        st = hip.resident_stats()
        assert st["bytes"] <= 5_500_000
        got_a = hip.matmul(a, z)                # a again: must not see b's transpose
        assert_equal(got_a, want_a, tol=1e-9, atol=1e-11)
        assert_equal(hip.matmul(b, zb), want_b, tol=1e-9, atol=1e-11)
        assert hip.resident_stats()["bytes"] <= 5_500_000
        hip.resident_clear()
        assert hip.resident_stats() ["entries"] == 0 and hip.resident_stats()["bytes"] == 0
    finally:
        hip.resident_set_limit(0)
