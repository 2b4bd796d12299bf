"""HIP path vs the CPU oracle on the same seeded inputs, at sizes the oracle
finishes in seconds (randomSparseArray()-style inputs plus special values)."""
import numpy as np
import pytest

from helpers import assert_equal, assert_identical, random_csc
from sparsearray_amd import NA_integer, NA_real, SVT_SparseArray

pytestmark = pytest.mark.gpu


def _svt(nrow, ncol, density, seed, dtype="double"):
    cp, ri, v = random_csc(nrow, ncol, density, seed, dtype)
    t = "double" if dtype == "double" else "integer"
    return SVT_SparseArray.from_csc((nrow, ncol), t, cp, ri, v)


def _sprinkle(x, seed, what):
    """Plant special values into some leaves (copies the value arrays)."""
    rng = np.random.default_rng(seed)
    leaves = []
    for lf in x.leaves:
        if lf is None or rng.random() > 0.3:
            leaves.append(lf)
            continue
        offs, vals = lf
        vals = vals.copy()
        vals[rng.integers(0, len(vals))] = what[rng.integers(0, len(what))]
        leaves.append((offs, vals))
    return SVT_SparseArray(x.dim, x.type, leaves)


SPECIAL_D = [NA_real, np.nan, np.inf, -np.inf]


def test_crossprod_large_takes_panel_kernels(hip, oracle):
    """nnz * K >= 2^28: the host-level entry points build the panel-blocked layout
    and run the LDS-DMA kernel (svt_hip.cpp, dev_crossprod_chunked); row-split
    partial sums, so the bar is the tolerance, not bit identity.  A NaN in the
    dense operand sends the same call through the general kernels on the device."""
    x = _svt(300_000, 2000, 0.01, 5)                    # 6e6 nonzeros
    x = _sprinkle(x, 6, [NA_real])                      # some leaves hold an R NA
    rng = np.random.default_rng(7)
    y = rng.uniform(-1, 1, (300_000, 70))               # K = 70: a partial dense tile too
    assert_equal(hip.crossprod(x, y), oracle.crossprod(x, y), tol=1e-9, atol=1e-11, strict_na=True)
    assert_equal(hip.crossprod(y, x), oracle.crossprod(y, x), tol=1e-9, atol=1e-11, strict_na=True)
    y[123_456, 3] = np.nan
    assert_equal(hip.crossprod(x, y), oracle.crossprod(x, y), tol=1e-9, atol=1e-11, strict_na=True)
    # sparse x sparse through the same kernels (the other operand densified in chunks)
    z = _svt(300_000, 60, 0.02, 8)      # (clean: a dirty leaf costs the oracle a full-length walk per dot)
    assert_equal(hip.crossprod(x, z), oracle.crossprod(x, z), tol=1e-9, atol=1e-11, strict_na=True)


@pytest.mark.parametrize("seed", [1, 2])
@pytest.mark.parametrize("K", [1, 7, 64, 130])
def test_crossprod_svt_dense(hip, oracle, seed, K):
    x = _svt(3000, 257, 0.01, seed)
    y = np.random.default_rng(seed + 10).uniform(-1, 1, (3000, K))
    assert_identical(hip.crossprod(x, y), oracle.crossprod(x, y))
    assert_identical(hip.crossprod(y, x), oracle.crossprod(y, x))
    yt = np.asfortranarray(y.T)
    assert_identical(hip.tcrossprod(x.t(), yt), oracle.tcrossprod(x.t(), yt))


@pytest.mark.parametrize("seed", [3, 4])
def test_crossprod_special_values(hip, oracle, seed):
    x = _sprinkle(_svt(500, 40, 0.05, seed), seed, SPECIAL_D)
    rng = np.random.default_rng(seed)
    y = rng.uniform(-1, 1, (500, 9))
    y[rng.integers(0, 500, 6), rng.integers(0, 9, 6)] = [np.inf, -np.inf, np.nan, NA_real, np.inf, np.nan]
    for a, b in ((x, y), (y, x)):
        assert_equal(hip.crossprod(a, b), oracle.crossprod(a, b), tol=1e-12, strict_na=True)
    assert_equal(hip.crossprod(x, x), oracle.crossprod(x, x), tol=1e-12, strict_na=True)
    assert_equal(hip.crossprod(x), oracle.crossprod(x), tol=1e-12, strict_na=True)


@pytest.mark.parametrize("seed", [5])
def test_crossprod_int(hip, oracle, seed):
    x = _svt(800, 33, 0.03, seed, "int")
    xn = _sprinkle(x, seed, [NA_integer])
    rng = np.random.default_rng(seed)
    y = rng.integers(-9, 9, (800, 5)).astype(np.int32)
    yn = y.copy()
    yn[3, 2] = NA_integer
    for a in (x, xn):
        for b in (y, yn):
            assert_identical(hip.crossprod(a, b), oracle.crossprod(a, b))
            assert_identical(hip.crossprod(b, a), oracle.crossprod(b, a))
        assert_identical(hip.crossprod(a), oracle.crossprod(a))
        assert_identical(hip.crossprod(a, xn), oracle.crossprod(a, xn))


@pytest.mark.parametrize("seed", [6, 7])
def test_sparse_sparse_and_matmul(hip, oracle, seed):
    a = _svt(2500, 40, 0.07, seed)
    b = _svt(2500, 65, 0.2, seed + 1)
    assert_identical(hip.crossprod(a, b), oracle.crossprod(a, b))
    assert_identical(hip.crossprod(b, a), oracle.crossprod(b, a))
    got, want = hip.crossprod(a), oracle.crossprod(a)
    assert_identical(got, want)
    assert np.array_equal(got, got.T)
    c = _svt(40, 12, 0.3, seed + 2)
    assert_identical(hip.matmul(a, c), oracle.matmul(a, c))


def _transposed(x):
    """t(x) of a 2-d SVT through scipy (the host mirror's own t() walks element by element)."""
    import scipy.sparse as sp
    cp = np.zeros(x.dim[1] + 1, dtype=np.int64)
    ri, vv = [], []
    for j, lf in enumerate(x.leaves):
        n = 0 if lf is None else len(lf[0])
        cp[j + 1] = cp[j] + n
        if n:
            ri.append(lf[0]); vv.append(lf[1])
    m = sp.csc_matrix((np.concatenate(vv), np.concatenate(ri), cp), shape=x.dim)
    t = m.T.tocsc()
    t.sort_indices()
    return SVT_SparseArray.from_csc((x.dim[1], x.dim[0]), x.type, t.indptr.astype(np.int64),
                                    t.indices.astype(np.int32), t.data)


@pytest.mark.parametrize("seed", [8, 9])
def test_matmul_one_call(hip, oracle, seed):
    """x %*% y through svt_matmul_SVT_{mat,SVT}: x is transposed on the device inside the
    call; the reference does t(x) on the host, then the crossprod2 entry points."""
    x = _sprinkle(_svt(700, 300, 0.03, seed), seed, SPECIAL_D)       # 700 x 300
    rng = np.random.default_rng(seed)
    y = rng.uniform(-1, 1, (300, 11))
    assert_equal(hip.matmul(x, y), oracle.matmul(x, y), tol=1e-12, strict_na=True)
    y[rng.integers(0, 300, 3), rng.integers(0, 11, 3)] = [np.inf, np.nan, NA_real]
    assert_equal(hip.matmul(x, y), oracle.matmul(x, y), tol=1e-12, strict_na=True)
    b = _svt(300, 40, 0.1, seed + 1)
    # (a row of x holding both an NA and a NaN: which payload the sum keeps depends on the
    # operand order of each addition, which neither R nor IEEE 754 pins -- NaN-ness must agree)
    assert_equal(hip.matmul(x, b), oracle.matmul(x, b), tol=1e-12)
    xi = _svt(700, 300, 0.03, seed + 2, "int")
    bi = _sprinkle(_svt(300, 9, 0.2, seed + 3, "int"), seed, [NA_integer])
    assert_identical(hip.matmul(xi, bi), oracle.matmul(xi, bi))
    yi = rng.integers(-5, 6, (300, 4)).astype(np.int32)
    assert_identical(hip.matmul(xi, yi), oracle.matmul(xi, yi))
    assert_equal(hip.matmul(xi, y), oracle.matmul(xi, y), tol=1e-12, strict_na=True)   # int x double
    from sparsearray_amd import SparseArrayError
    with pytest.raises(SparseArrayError, match="non-conformable"):
        hip.matmul(x, np.zeros((299, 2)))
    with pytest.raises(SparseArrayError, match="non-conformable"):
        hip.matmul(x, _svt(299, 4, 0.1, 1))
    e = SVT_SparseArray((700, 300), "double", [None] * 300)
    assert not hip.matmul(e, y).any()


def test_matmul_one_call_large(hip, oracle):
    """nnz * K >= 2^28: transposition + panel-blocked layout + LDS-DMA kernel in one call."""
    xt = _svt(2000, 300_000, 0.01, 21)          # t(x): what crossprod() would be given
    x = _transposed(xt)                          # 300000 x 2000, 6e6 nonzeros
    rng = np.random.default_rng(22)
    y = rng.uniform(-1, 1, (2000, 70))
    want = oracle.crossprod(xt, y)               # = x %*% y
    assert_equal(hip.matmul(x, y), want, tol=1e-9, atol=1e-11)
    b = _svt(2000, 50, 0.05, 23)
    assert_equal(hip.matmul(x, b), oracle.crossprod(xt, b), tol=1e-9, atol=1e-11)


OPS_COL = ["colSums", "colMeans", "colVars", "colSds", "colMins", "colMaxs",
           "colProds", "colAnyNAs", "colCountNAs"]
OPS_ROW = ["rowSums", "rowMeans", "rowVars", "rowSds", "rowMins", "rowMaxs",
           "rowAnyNAs", "rowCountNAs"]


@pytest.mark.parametrize("na_rm", [False, True])
@pytest.mark.parametrize("shape,density", [((1000, 300), 0.01), ((20000, 12), 0.2),
                                           ((60, 50, 8), 0.05)])
def test_matrixstats_double(hip, oracle, shape, density, na_rm):
    ncol = int(np.prod(shape[1:]))
    x2 = _sprinkle(_svt(shape[0], ncol, density, 11), 11, SPECIAL_D[:2] + [3.5, -2.0])
    x = SVT_SparseArray(shape, "double", x2.leaves)
    for dims in range(1, len(shape)):
        for op in OPS_COL + OPS_ROW:
            kw = {} if "AnyNAs" in op or "CountNAs" in op else {"na_rm": na_rm}
            got = getattr(hip, op)(x, dims=dims, **kw)
            want = getattr(oracle, op)(x, dims=dims, **kw)
            if got.dtype == np.int32:
                assert_identical(got, want, op)
            else:
                assert_equal(got, want, tol=1e-6, what=f"{op} dims={dims}",
                             strict_na=op[3:] in ("Mins", "Maxs"), atol=1e-9)


@pytest.mark.parametrize("na_rm", [False, True])
@pytest.mark.parametrize("shape,density,dtype", [((40000, 3000), 0.01, "double"), ((40000, 2000), 0.01, "int"),
                                                 ((20000, 50, 40), 0.02, "double"), ((16384, 700), 0.05, "double"),
                                                 ((16385, 9), 0.5, "int"),
                                                 # many output columns of few short leaves, all rows in LDS
                                                 # (rowstats_whole_kernel, dims = 2)
                                                 ((9000, 1100, 6), 0.01, "double"), ((8500, 1030, 3), 0.02, "int")])
def test_row_stats_long_panels_and_strata_ranges(hip, oracle, shape, density, dtype, na_rm):
    """>= 16384 rows: the sum-like row statistics take 8192-row panels and, with few panels, cut
    the leaves into ranges whose partial cells are added in `out` (kernels_rowstats.hip);
    rowMins / rowMaxs keep the short panels.  src/SparseArray_matrixStats.c:599-696."""
    ncol = int(np.prod(shape[1:]))
    special = SPECIAL_D[:2] + [3.5, -2.0] if dtype == "double" else [NA_integer, -7, 12]
    x2 = _sprinkle(_svt(shape[0], ncol, density, 21, dtype), 21, special)
    x = SVT_SparseArray(shape, x2.type, x2.leaves)
    for dims in range(1, len(shape)):
        for op in OPS_ROW:
            kw = {} if "AnyNAs" in op or "CountNAs" in op else {"na_rm": na_rm}
            got = getattr(hip, op)(x, dims=dims, **kw)
            want = getattr(oracle, op)(x, dims=dims, **kw)
            if got.dtype == np.int32 or (dtype == "int" and op in ("rowSums", "rowCountNAs")):
                assert_identical(got, want, f"{op} dims={dims}")
            else:
                assert_equal(got, want, tol=1e-9, what=f"{op} dims={dims}",
                             strict_na=op[3:] in ("Mins", "Maxs"), atol=1e-9)


@pytest.mark.parametrize("na_rm", [False, True])
@pytest.mark.parametrize("shape,fill,type_", [((3000, 200), 0.9, "double"), ((500, 40, 6), 0.3, "double"),
                                              ((4000, 64), 0.6, "integer")])
def test_matrixstats_NaArray_col_ops(hip, oracle, shape, fill, type_, na_rm):
    """NaArray operands (NA background, R/NaArray-matrixStats.R): column statistics and
    whole-array summaries; some columns have no NA at all, some nothing but NAs."""
    rng = np.random.default_rng(17)
    if type_ == "double":
        a = np.round(rng.normal(size=shape), 2)
        a[rng.random(shape) < 0.02] = np.nan                # stored NaNs next to the NA background
        a[rng.random(shape) > fill] = NA_real
        a[:, 0] = np.round(rng.normal(size=a[:, 0].shape), 2)    # complete
        a[:, 1] = NA_real                                         # empty
    else:
        a = rng.integers(-20, 20, shape).astype(np.int32)
        a[rng.random(shape) > fill] = NA_integer
        a[:, 0] = 3
        a[:, 1] = NA_integer
    x = SVT_SparseArray.from_dense(np.asfortranarray(a), type_, na_background=True)
    ops = ["colSums", "colMeans", "colVars", "colSds", "colMins", "colMaxs", "colProds",
           "colAnyNAs", "colCountNAs"] + (["colAnys", "colAlls"] if type_ == "integer" else [])
    import warnings
    for dims in range(1, len(shape)):
        for op in ops:
            kw = {} if "AnyNAs" in op or "CountNAs" in op else {"na_rm": na_rm}
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                got = getattr(hip, op)(x, dims=dims, **kw)
                want = getattr(oracle, op)(x, dims=dims, **kw)
            if got.dtype == np.int32:
                assert_identical(got, want, op)
            else:
                assert_equal(got, want, tol=1e-6, what=f"{op} dims={dims}", atol=1e-9,
                             strict_na=op[3:] in ("Mins", "Maxs"))
    for op in ["sum", "mean", "min", "max", "var", "anyNA"]:
        kw = {} if op == "anyNA" else {"na_rm": na_rm}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got, want = getattr(hip, op)(x, **kw), getattr(oracle, op)(x, **kw)
        assert_equal(np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64), tol=1e-6, atol=1e-9, what=op)
    # row statistics the reference defines for NaArray objects
    for dims in range(1, len(shape)):
        for op in ["rowSums", "rowMins", "rowMaxs", "rowAnyNAs", "rowCountNAs"]:
            kw = {} if "AnyNAs" in op or "CountNAs" in op else {"na_rm": na_rm}
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                got = getattr(hip, op)(x, dims=dims, **kw)
                want = getattr(oracle, op)(x, dims=dims, **kw)
            if got.dtype == np.int32:
                assert_identical(got, want, op)
            else:
                # NA + NaN in one cell: which payload an IEEE sum keeps is not pinned
                # (SURVEY.md section 8a notes); min/max return NA_real_ explicitly
                assert_equal(got, want, tol=1e-6, what=f"{op} dims={dims}", atol=1e-9,
                             strict_na=op[3:] in ("Mins", "Maxs"))
    for sess in (hip, oracle):
        with pytest.raises(Exception, match="NaArray"):
            sess.rowMeans(x)
        with pytest.raises(Exception, match="NaArray"):
            sess.rowVars(x)
        if len(shape) == 2:
            with pytest.raises(Exception, match="NaMatrix"):
                sess.crossprod(x, np.ones((shape[0], 2)))


@pytest.mark.parametrize("na_rm", [False, True])
@pytest.mark.parametrize("type_", ["double", "integer", "NaArray"])
def test_colstats_few_long_segments(hip, oracle, type_, na_rm):
    """A handful of very long generalized columns and whole-array summaries: the
    device cuts each segment into chunks (kernels_colstats.hip, launch_colstats_split)
    and combines the partial states; same answers as one pass per segment."""
    import warnings
    rng = np.random.default_rng(23)
    shape = (180_000, 3, 2)
    if type_ == "integer":
        a = rng.integers(-9, 10, shape).astype(np.int32)
        a[rng.random(shape) < 0.4] = 0
        a[77, 1, 0] = NA_integer
        a[:, 2, 1] = np.abs(a[:, 2, 1]) + 1        # a column without zeros (colAlls)
        x = SVT_SparseArray.from_dense(np.asfortranarray(a), "integer")
    else:
        a = np.round(rng.normal(size=shape), 3)
        if type_ == "NaArray":
            a[rng.random(shape) < 0.3] = NA_real
            a[:, 0, 1] = np.round(rng.normal(size=shape[0]), 3)      # complete column
        else:
            a[rng.random(shape) < 0.4] = 0.0
            a[170_000, 1, 0] = NA_real
        a[5, 2, 0] = np.nan
        a[9, 0, 0] = np.inf
        x = SVT_SparseArray.from_dense(np.asfortranarray(a), "double", na_background=type_ == "NaArray")
    ops = OPS_COL + (["colAnys", "colAlls"] if type_ == "integer" else [])
    for dims in (1, 2):
        for op in ops:
            kw = {} if "AnyNAs" in op or "CountNAs" in op else {"na_rm": na_rm}
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                got = getattr(hip, op)(x, dims=dims, **kw)
                want = getattr(oracle, op)(x, dims=dims, **kw)
            if got.dtype == np.int32:
                assert_identical(got, want, op)
            else:
                assert_equal(got, want, tol=1e-9, what=f"{op} dims={dims}", atol=1e-9,
                             strict_na=op[3:] in ("Mins", "Maxs"))
    sops = ["sum", "mean", "min", "max", "range", "anyNA"] + \
        (["any", "all"] if type_ == "integer" else ["var", "sd", "prod"])
    for op in sops:
        kw = {} if op == "anyNA" else {"na_rm": na_rm}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got, want = getattr(hip, op)(x, **kw), getattr(oracle, op)(x, **kw)
        assert_equal(np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64),
                     tol=1e-9, atol=1e-9, what=op)


@pytest.mark.parametrize("na_rm", [False, True])
def test_matrixstats_int_exact(hip, oracle, na_rm):
    x = _sprinkle(_svt(5000, 64, 0.05, 12, "int"), 12, [NA_integer, -7])
    for op in ["colSums", "colMeans", "colMins", "colMaxs", "colAnys", "colAlls",
               "rowSums", "rowMins", "rowMaxs", "rowAnys", "rowAlls", "colVars"]:
        got = getattr(hip, op)(x, na_rm=na_rm)
        want = getattr(oracle, op)(x, na_rm=na_rm)
        if op == "colVars":
            assert_equal(got, want, tol=1e-9, what=op, strict_na=True)
        else:
            assert_identical(got, want, op)   # integer work: bit-exact


@pytest.mark.parametrize("ngroup", [3, 1000])
@pytest.mark.parametrize("na_rm", [False, True])
def test_rowsum_colsum(hip, oracle, ngroup, na_rm):
    rng = np.random.default_rng(13)
    x = _sprinkle(_svt(6000, 50, 0.05, 13), 13, SPECIAL_D)
    grp = list(rng.integers(0, ngroup, 6000))
    got, ug1 = hip.rowsum(x, grp, na_rm=na_rm)
    want, ug2 = oracle.rowsum(x, grp, na_rm=na_rm)
    assert ug1 == ug2
    assert_equal(got, want, tol=1e-9, atol=1e-10)
    xt = x.t()
    got, _ = hip.colsum(xt, grp, na_rm=na_rm)
    want, _ = oracle.colsum(xt, grp, na_rm=na_rm)
    assert_equal(got, want, tol=1e-9, atol=1e-10)
    xi = _sprinkle(_svt(6000, 50, 0.05, 14, "int"), 14, [NA_integer])
    got, _ = hip.rowsum(xi, grp, na_rm=na_rm)
    want, _ = oracle.rowsum(xi, grp, na_rm=na_rm)
    assert_identical(got, want)
    got, _ = hip.colsum(xi.t(), grp, na_rm=na_rm)
    want, _ = oracle.colsum(xi.t(), grp, na_rm=na_rm)
    assert_identical(got, want)


def test_summarize(hip, oracle):
    x = _svt(4000, 100, 0.02, 15)
    for op in ["sum", "mean", "var", "sd", "min", "max", "range", "prod", "anyNA"]:
        assert_equal(getattr(hip, op)(x), getattr(oracle, op)(x), tol=1e-6, what=op, atol=1e-9)
    xi = _svt(4000, 100, 0.02, 16, "int")
    for op in ["sum", "min", "max", "range", "any", "all", "anyNA"]:
        assert_identical(getattr(hip, op)(xi), getattr(oracle, op)(xi), op)


def test_error_behaviour(hip):
    from sparsearray_amd import SparseArrayError
    x = _svt(100, 10, 0.1, 17)
    with pytest.raises(SparseArrayError, match="non-conformable"):
        hip.crossprod(x, np.zeros((99, 3)))
    with pytest.raises(SparseArrayError, match="does not support"):
        hip.any(x)
    with pytest.raises(SparseArrayError, match="group"):
        hip.SparseArray_Call("C_rowsum_SVT", x, np.full(100, 7, np.int32), 3, False)


def test_resident_operands(hip, oracle):
    """svt_resident_set_limit(): later calls on the same object find its device copy (and
    its derived layouts); results are those of the uncached calls; LRU eviction; off again."""
    x = _sprinkle(_svt(20000, 400, 0.02, 31), 31, SPECIAL_D)
    x2 = _svt(20000, 300, 0.02, 32)
    xi = _svt(5000, 64, 0.05, 33, "int")
    rng = np.random.default_rng(34)
    y = rng.uniform(-1, 1, (20000, 5))
    z = rng.uniform(-1, 1, (400, 3))
    want = [hip.colSums(x), hip.colVars(x), hip.rowSums(x), hip.crossprod(x, y),
            hip.matmul(x, z), hip.matmul(x, z), hip.sum(x), hip.colSums(xi)]
    base = hip.resident_stats()
    assert base["entries"] == 0 and base["bytes"] == 0
    try:
        hip.resident_set_limit(1 << 30)
        got = [hip.colSums(x), hip.colVars(x), hip.rowSums(x), hip.crossprod(x, y),
               hip.matmul(x, z), hip.matmul(x, z), hip.sum(x), hip.colSums(xi)]
        for g, w in zip(got, want):          # (rowSums adds in LDS-atomic order: not bit-reproducible)
            assert_equal(np.asarray(g, dtype=np.float64), np.asarray(w, dtype=np.float64),
                         tol=1e-12, atol=1e-13, strict_na=True)
        st = hip.resident_stats()
        assert st["entries"] == 2                           # x (with t(x)) and xi
        assert st["hits"] - base["hits"] == 6 and st["misses"] - base["misses"] == 2
        assert st["bytes"] >= 2 * 12 * 150_000             # x and t(x)
        # a different object with equal shape is a different operand
        assert_equal(hip.colSums(x2), oracle.colSums(x2), tol=1e-12)
        assert hip.resident_stats()["entries"] == 3
        # a modified copy of x (new leaf arrays) is not mistaken for x
        leaves = list(x.leaves)
        offs, vals = leaves[7]
        leaves[7] = (offs.copy(), vals.copy() * 2.0)
        xm = SVT_SparseArray(x.dim, x.type, leaves)
        assert_equal(hip.colSums(xm), oracle.colSums(xm), tol=1e-12, strict_na=True)
        # LRU: a limit that holds only the small operand drops the others
        hip.resident_set_limit(200_000)
        st = hip.resident_stats()
        assert st["bytes"] <= 200_000
        assert_identical(hip.colSums(x), want[0])             # too big to stay: one-call upload
        assert hip.resident_stats()["bytes"] <= 200_000
        hip.resident_clear()
        assert hip.resident_stats()["entries"] == 0
    finally:
        hip.resident_set_limit(0)
    assert hip.resident_stats()["bytes"] == 0
    assert_identical(hip.colSums(x), want[0])


@pytest.mark.parametrize("na_rm", [False, True])
@pytest.mark.parametrize("type_", ["double", "integer", "NaArray"])
def test_colstats_very_short_leaves(hip, oracle, type_, na_rm):
    """Thousands of leaves with a handful of nonzeros each (short first extent): one thread per
    generalized column on the device (kernels_colstats.hip, colstats_thread_kernel)."""
    import warnings
    rng = np.random.default_rng(29)
    shape = (6, 90, 60)
    if type_ == "integer":
        a = rng.integers(-9, 10, shape).astype(np.int32)
        a[rng.random(shape) < 0.7] = 0
        a[2, 5, 7] = NA_integer
        x = SVT_SparseArray.from_dense(np.asfortranarray(a), "integer")
    else:
        a = np.round(rng.normal(size=shape), 3)
        if type_ == "NaArray":
            a[rng.random(shape) < 0.7] = NA_real
        else:
            a[rng.random(shape) < 0.7] = 0.0
            a[1, 4, 4] = NA_real
        a[3, 8, 9] = np.nan
        a[0, 1, 2] = np.inf
        x = SVT_SparseArray.from_dense(np.asfortranarray(a), "double", na_background=type_ == "NaArray")
    ops = OPS_COL + (["colAnys", "colAlls"] if type_ == "integer" else [])
    for op in ops:
        kw = {} if "AnyNAs" in op or "CountNAs" in op else {"na_rm": na_rm}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = getattr(hip, op)(x, dims=1, **kw)
            want = getattr(oracle, op)(x, dims=1, **kw)
        if got.dtype == np.int32:
            assert_identical(got, want, op)
        else:
            assert_equal(got, want, tol=1e-12, what=op, atol=1e-12, strict_na=op[3:] in ("Mins", "Maxs"))


def test_crossprod_int_large_takes_panel_kernels(hip, oracle):
    """Integer operands with nnz * K >= 2^28 run on the f64 panel kernels through f64 copies made on
    the device (svt_hip.cpp, dev_crossprod_chunked / dev_crossprod_pp): integer arithmetic in double
    is what the reference does (src/SparseVec_dotprod.c:73-114), so the results are identical, the
    NA rules included."""
    x = _svt(200_000, 1500, 0.01, 71, "int")                       # 3e6 nonzeros
    rng = np.random.default_rng(72)
    y = rng.integers(-9, 10, (200_000, 96)).astype(np.int32)
    assert_identical(hip.crossprod(x, y), oracle.crossprod(x, y))
    xn = _sprinkle(x, 73, [NA_integer])                            # NA in some leaves
    assert_identical(hip.crossprod(xn, y), oracle.crossprod(xn, y))
    y[12_345, 7] = NA_integer                                      # NA in the dense operand
    assert_identical(hip.crossprod(xn, y), oracle.crossprod(xn, y))
    z = _svt(200_000, 100, 0.02, 74, "int")
    assert_identical(hip.crossprod(x, z), oracle.crossprod(x, z))


def test_crossprod_mixed_integer_double(hip, oracle):
    """Integer x double pairs: the R methods coerce the integer operand on the host first; the HIP
    library takes the pair as it is and widens on the device (include/svt_hip.h).  Same results as
    the coerced call, small (general kernels) and large (panel kernels)."""
    rng = np.random.default_rng(81)
    for nrow, ncol, K in ((900, 70, 9), (60_000, 1000, 224)):
        xi = _svt(nrow, ncol, 0.02, 82, "int")          # large case: 1.2e6 nonzeros x 224 >= 2^28
        xd = _svt(nrow, ncol, 0.02, 84)
        yd = rng.uniform(-1, 1, (nrow, K))
        yi = rng.integers(-5, 6, (nrow, K)).astype(np.int32)
        if nrow < 1000:       # (NAs make the oracle walk whole columns: small case only)
            xi = _sprinkle(xi, 83, [NA_integer])
            yi[7, 2] = NA_integer
        tol = dict(tol=1e-9, atol=1e-10, strict_na=True)
        assert_equal(hip.crossprod(xi, yd), oracle.crossprod(xi, yd), **tol)
        assert_equal(hip.crossprod(yd, xi), oracle.crossprod(yd, xi), **tol)
        assert_equal(hip.crossprod(xd, yi), oracle.crossprod(xd, yi), **tol)
        assert_equal(hip.crossprod(yi, xd), oracle.crossprod(yi, xd), **tol)
    zi = _svt(70, 900, 0.05, 85, "int")
    w = rng.uniform(-1, 1, (900, 4))
    assert_equal(hip.matmul(zi, w), oracle.matmul(zi, w), tol=1e-12, strict_na=True)


def test_crossprod_wide_dense_operand_is_chunked(hip, oracle):
    """More than 512 dense columns: the host entry point runs the panel kernels chunk by chunk
    (svt_hip.cpp, dev_crossprod_chunked); both orientations."""
    x = _svt(60_000, 1000, 0.01, 91)
    y = np.random.default_rng(92).uniform(-1, 1, (60_000, 600))
    want = oracle.crossprod(x, y)
    assert_equal(hip.crossprod(x, y), want, tol=1e-9, atol=1e-11)
    assert_equal(hip.crossprod(y, x), want.T, tol=1e-9, atol=1e-11)


def test_crossprod1_large_is_triangular_and_symmetric(hip, oracle):
    """Unary crossprod(x) above the panel-kernel threshold: every dense chunk of x's own columns is
    multiplied with the leaves from its first column on only (src/SparseMatrix_mult.c:263-296) and
    the result is mirrored.  An NA in one leaf, an Inf in another (a dirty dense column in its own
    chunk) and a second chunk (ncol > 512)."""
    x = _svt(150_000, 1300, 0.01, 91)
    leaves = list(x.leaves)
    offs, vals = leaves[700]
    vals = vals.copy(); vals[5] = np.inf
    leaves[700] = (offs, vals)
    offs, vals = leaves[3]
    vals = vals.copy(); vals[0] = NA_real
    leaves[3] = (offs, vals)
    x = SVT_SparseArray(x.dim, x.type, leaves)
    got = hip.crossprod(x)
    want = oracle.crossprod(x)
    assert_equal(got, want, tol=1e-9, atol=1e-11, strict_na=True, what="crossprod(x)")
    g = np.asarray(got)
    assert np.array_equal(g, g.T, equal_nan=True)


def test_crossprod_large_very_sparse_takes_gather_kernel(hip, oracle):
    """Host-level crossprod above the panel-kernel threshold at 0.1 % density: the layout chosen by density is
    the gather one (rows of Y straight from L2).  Both operand orders, the transposed orientation, an NA leaf
    and a non-finite entry in the dense operand."""
    x = _svt(2_000_000, 2000, 0.001, 97)                # 4e6 nonzeros, K = 70: nnz * K = 2.8e8
    x = _sprinkle(x, 98, [NA_real])
    rng = np.random.default_rng(99)
    y = rng.uniform(-1, 1, (2_000_000, 70))
    assert_equal(hip.crossprod(x, y), oracle.crossprod(x, y), tol=1e-9, atol=1e-11, strict_na=True)
    assert_equal(hip.crossprod(y, x), oracle.crossprod(y, x), tol=1e-9, atol=1e-11, strict_na=True)
    yt = np.asfortranarray(y.T)
    assert_equal(hip.tcrossprod(x.t(), yt), oracle.tcrossprod(x.t(), yt), tol=1e-9, atol=1e-11, strict_na=True)
    y[1_234_567, 3] = np.inf
    assert_equal(hip.crossprod(x, y), oracle.crossprod(x, y), tol=1e-9, atol=1e-11, strict_na=True)
