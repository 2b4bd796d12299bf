"""Round-3 parity hardening ("Next round" item 7).

* BASELINE config 1 at its exact shape (randomSparseArray(c(1e4, 1e3), density=0.01): 1e5
  nonzeros) as a ``-m gpu`` test: colSums / colVars / the other column statistics against the
  oracle, integer and logical results bit for bit.
* HIP vs oracle through the RAW dispatchers -- the two C ABIs called with byte-identical
  arguments, nothing of sparsearray_amd/api.py (type promotion, rowVars from sums, group
  matching) in between, so logic shared by both sessions cannot mask a difference.
* The dgCMatrix column statistics and t() through the C ABI at sizes past the golden cases.
"""
import numpy as np
import pytest
import torch  # noqa: F401  (before the HIP library: one HIP runtime per process, see sparsearray_amd/_hip.py)

from helpers import assert_equal, assert_identical, random_csc
from sparsearray_amd import NA_integer, NA_real, SVT_SparseArray

pytestmark = pytest.mark.gpu


def _config1(dtype="double"):
    cp, ri, v = random_csc(10_000, 1_000, 0.01, seed=1, dtype=dtype)
    assert cp[-1] == 100_000                     # R/randomSparseArray.R:25-26: floor(prod(dim) * density)
    t = "double" if dtype == "double" else "integer"
    return SVT_SparseArray.from_csc((10_000, 1_000), t, cp, ri, v)


def test_config1_exact_shape_double(hip, oracle):
    """The device adds a leaf's ~100 values with 16 lanes (a different order from the oracle's
    sequential loop): the doubles agree to 1e-12, far inside the 1e-6 bar; every statistic whose
    result does not depend on the order of additions is bit-identical."""
    x = _config1()
    for fn in ("colSums", "colMeans", "colVars", "colSds"):
        assert_equal(getattr(hip, fn)(x), getattr(oracle, fn)(x), tol=1e-12, atol=1e-15, what=fn)
    for fn in ("colMins", "colMaxs", "colAnyNAs", "colCountNAs"):
        assert_identical(getattr(hip, fn)(x), getattr(oracle, fn)(x), fn)
    assert_equal(hip.sum(x), oracle.sum(x), tol=1e-12, what="sum")
    assert_equal(hip.rowSums(x), oracle.rowSums(x), tol=1e-12, atol=1e-15, what="rowSums")


def test_config1_exact_shape_integer_bit_exact(hip, oracle):
    """Integer input: sums accumulate in double and every partial sum is an exact integer
    (src/Rvector_summarization.c:518-537), so the results are bit-identical whatever the order."""
    x = _config1("int")
    for fn in ("colSums", "colMins", "colMaxs", "colAnys", "colAlls", "colMeans", "rowSums"):
        assert_identical(getattr(hip, fn)(x), getattr(oracle, fn)(x), fn)
    assert_equal(hip.colVars(x), oracle.colVars(x), tol=1e-12, what="colVars")


# ---------------------------------------------------------------------------------------------
# raw dispatcher calls: no api.Session in between
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def raw():
    from oracle.oracle import oracle_dispatcher
    from sparsearray_amd._hip import hip_dispatcher
    return hip_dispatcher(), oracle_dispatcher()


def _special(nrow, ncol, density, seed, dtype="double", one_per_leaf=False):
    """one_per_leaf: at most one special value per leaf.  A leaf that holds BOTH a NaN and an NA times a
    clean dense column goes through _dotprod_doubleSV_finite_doubles (src/SparseVec_dotprod.c:28-43: IEEE
    arithmetic only), whose NaN payload -- NA or not -- is whichever operand the hardware propagates;
    neither the reference's tests nor the parity rule pin that ("R NA vs NaN")."""
    cp, ri, v = random_csc(nrow, ncol, density, seed, dtype)
    rng = np.random.default_rng(seed + 100)
    v = v.copy()
    if one_per_leaf:
        lens = np.diff(cp)
        leaves = np.flatnonzero((lens > 0) & (rng.random(ncol) < 0.4))
        pick = cp[leaves] + (rng.random(len(leaves)) * lens[leaves]).astype(np.int64)
    else:
        pick = rng.choice(len(v), size=max(4, len(v) // 50), replace=False)
    if dtype == "double":
        v[pick] = rng.choice([NA_real, np.nan, np.inf, -np.inf], size=len(pick))
    else:
        v[pick] = NA_integer
    t = "double" if dtype == "double" else "integer"
    return SVT_SparseArray.from_csc((nrow, ncol), t, cp, ri, v)


@pytest.mark.parametrize("dtype", ["double", "int"])
def test_raw_colstats_rowstats_summarize(raw, dtype):
    h, o = raw
    x = _special(700, 90, 0.04, 3, dtype)
    ops = ["anyNA", "countNAs", "min", "max", "sum", "prod", "mean", "var1", "sd1", "centered_X2_sum"]
    if dtype == "int":
        ops += ["any", "all"]
    for op in ops:
        for na_rm in (False, True):
            center = NA_real if op != "centered_X2_sum" else 0.25
            (a, wa), (b, wb) = h("C_colStats_SVT", x, op, na_rm, center, 1), \
                o("C_colStats_SVT", x, op, na_rm, center, 1)
            assert wa == wb, (op, na_rm)
            if np.asarray(b).dtype == np.int32:
                assert_identical(a, b, f"colStats {op} na_rm={na_rm}")
            else:
                assert_equal(a, b, tol=1e-9, atol=1e-12, strict_na=op in ("min", "max", "sum", "mean", "prod"),
                             what=f"colStats {op} na_rm={na_rm}")
            (a, wa), (b, wb) = h("C_summarize_SVT", x, op, na_rm, center), o("C_summarize_SVT", x, op, na_rm, center)
            assert wa == wb, (op, na_rm)
            if np.asarray(b).dtype == np.int32:
                assert_identical(a, b, f"summarize {op} na_rm={na_rm}")
            else:
                assert_equal(a, b, tol=1e-9, atol=1e-12, what=f"summarize {op} na_rm={na_rm}")
    for op in ("countNAs", "anyNA", "min", "max", "sum", "centered_X2_sum"):
        for na_rm in (False, True):
            center = None if op != "centered_X2_sum" else np.linspace(-1, 1, 700)
            (a, wa), (b, wb) = h("C_rowStats_SVT", x, op, na_rm, center, 1), \
                o("C_rowStats_SVT", x, op, na_rm, center, 1)
            assert wa == wb, (op, na_rm)
            if np.asarray(b).dtype == np.int32:
                assert_identical(a, b, f"rowStats {op} na_rm={na_rm}")
            else:
                assert_equal(a, b, tol=1e-9, atol=1e-12, strict_na=op in ("min", "max", "sum"),
                             what=f"rowStats {op} na_rm={na_rm}")


def test_raw_crossprod_entry_points(raw):
    h, o = raw
    x = _special(600, 37, 0.05, 11, one_per_leaf=True)
    y = np.asfortranarray(np.random.default_rng(12).uniform(-1, 1, (600, 19)))
    y[5, 2] = np.inf
    y[77, 4] = NA_real
    z = _special(600, 23, 0.06, 13, one_per_leaf=True)
    assert_equal(h("C_crossprod2_SVT_mat", x, y, False), o("C_crossprod2_SVT_mat", x, y, False),
                 tol=1e-12, strict_na=True, what="SVT_mat")
    yt = np.asfortranarray(y.T)
    assert_equal(h("C_crossprod2_SVT_mat", x, yt, True), o("C_crossprod2_SVT_mat", x, yt, True),
                 tol=1e-12, strict_na=True, what="SVT_mat tr_y")
    assert_equal(h("C_crossprod2_mat_SVT", y, x, False), o("C_crossprod2_mat_SVT", y, x, False),
                 tol=1e-12, strict_na=True, what="mat_SVT")
    assert_equal(h("C_crossprod2_mat_SVT", yt, x, True), o("C_crossprod2_mat_SVT", yt, x, True),
                 tol=1e-12, strict_na=True, what="mat_SVT tr_x")
    assert_equal(h("C_crossprod2_SVT_SVT", x, z), o("C_crossprod2_SVT_SVT", x, z),
                 tol=1e-12, strict_na=True, what="SVT_SVT")
    assert_equal(h("C_crossprod1_SVT", x), o("C_crossprod1_SVT", x), tol=1e-12, strict_na=True, what="crossprod1")


@pytest.mark.parametrize("dtype", ["double", "int"])
def test_raw_rowsum_colsum(raw, dtype):
    h, o = raw
    x = _special(500, 60, 0.05, 21, dtype)
    rng = np.random.default_rng(22)
    g = rng.integers(1, 8, 500).astype(np.int32)
    g[rng.integers(0, 500, 5)] = NA_integer                 # NA group -> the last group
    for na_rm in (False, True):
        (a, ova), (b, ovb) = h("C_rowsum_SVT", x, g, 8, na_rm), o("C_rowsum_SVT", x, g, 8, na_rm)
        assert ova == ovb
        (assert_identical if dtype == "int" else
         (lambda p, q, w: assert_equal(p, q, tol=1e-12, atol=1e-14, what=w)))(a, b, f"rowsum na_rm={na_rm}")
        gc = rng.integers(1, 6, 60).astype(np.int32)
        (a, ova), (b, ovb) = h("C_colsum_SVT", x, gc, 5, na_rm), o("C_colsum_SVT", x, gc, 5, na_rm)
        assert ova == ovb
        (assert_identical if dtype == "int" else
         (lambda p, q, w: assert_equal(p, q, tol=1e-12, atol=1e-14, what=w)))(a, b, f"colsum na_rm={na_rm}")


def test_raw_transpose_and_aperm(raw):
    h, o = raw
    for dtype in ("double", "int"):
        x = _special(300, 45, 0.07, 31, dtype)
        a, b = h("C_transpose_2D_SVT", x), o("C_transpose_2D_SVT", x)
        assert a.dim == b.dim == (45, 300)
        assert_identical(a.to_dense(), b.to_dense(), f"t() {dtype}")
        for la, lb in zip(a.leaves, b.leaves):               # leaf by leaf: same offsets in the same order
            assert (la is None) == (lb is None)
            if la is not None:
                assert np.array_equal(la[0], lb[0])
        a2, b2 = h("C_aperm_SVT", x, [2, 1]), o("C_aperm_SVT", x, [2, 1])
        assert_identical(a2.to_dense(), b2.to_dense(), f"aperm(2,1) {dtype}")
        assert_identical(a2.to_dense(), a.to_dense(), "aperm(2,1) == t()")


# ---------------------------------------------------------------------------------------------
# dgCMatrix column statistics (src/sparseMatrix_utils.c:105-223) past the golden sizes
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(5000, 300, 0.02), (64, 2000, 0.3), (3, 17, 0.9), (1, 9, 1.0)])
def test_dgCMatrix_column_statistics(hip, oracle, shape):
    nrow, ncol, dens = shape
    cp, ri, v = random_csc(nrow, ncol, dens, seed=41)
    v = v.copy()
    rng = np.random.default_rng(42)
    if len(v) > 8:
        pick = rng.choice(len(v), size=max(3, len(v) // 40), replace=False)
        v[pick] = rng.choice([NA_real, np.nan, np.inf, -np.inf, 0.0], size=len(pick))   # explicit zeros are legal
    g = ((nrow, ncol), cp.astype(np.int32), ri, v)
    for na_rm in (False, True):
        for fn in ("colMins_dgCMatrix", "colMaxs_dgCMatrix", "colRanges_dgCMatrix"):
            assert_identical(getattr(hip, fn)(g, na_rm), getattr(oracle, fn)(g, na_rm), f"{fn} na_rm={na_rm}")
        with np.errstate(all="ignore"):
            assert_equal(hip.colVars_dgCMatrix(g, na_rm), oracle.colVars_dgCMatrix(g, na_rm), tol=1e-9,
                         atol=1e-13, what=f"colVars_dgCMatrix na_rm={na_rm}")


def test_dgCMatrix_column_statistics_zero_extent(hip, oracle):
    for dim in ((0, 4), (5, 0), (0, 0)):
        g = (dim, np.zeros(dim[1] + 1, np.int32), np.zeros(0, np.int32), np.zeros(0))
        for fn in ("colMins_dgCMatrix", "colMaxs_dgCMatrix", "colRanges_dgCMatrix", "colVars_dgCMatrix"):
            with np.errstate(all="ignore"):
                a, b = getattr(hip, fn)(g, False), getattr(oracle, fn)(g, False)
            assert np.asarray(a).shape == np.asarray(b).shape
            assert_equal(a, b, tol=1e-12, what=f"{fn} {dim}")


def test_session_t_goes_through_the_device(hip, oracle):
    """tcrossprod / rowMedians / the non-native row statistics of a 2-D object transpose with ONE
    C_transpose_2D_SVT call (no host-side element loop): results as the oracle's."""
    x = _special(400, 50, 0.06, 51, one_per_leaf=True)
    y = np.asfortranarray(np.random.default_rng(52).uniform(-1, 1, (7, 50)))
    # (rows of x = leaves of t(x) may hold a NaN and an NA together: IEEE-propagated class, not pinned)
    assert_equal(hip.tcrossprod(x, y), oracle.tcrossprod(x, y), tol=1e-12, what="tcrossprod")
    assert_equal(hip.tcrossprod(x), oracle.tcrossprod(x), tol=1e-12, what="tcrossprod1")
    xi = _special(400, 50, 0.06, 53, "int")
    for fn in ("rowProds", "rowAnys", "rowAlls"):
        a, b = getattr(hip, fn)(xi), getattr(oracle, fn)(xi)
        if np.asarray(b).dtype == np.int32:
            assert_identical(a, b, fn)
        else:
            assert_equal(a, b, tol=1e-12, strict_na=True, what=fn)
    t1, t2 = hip.t(x), oracle.t(x)
    assert_identical(t1.to_dense(), t2.to_dense(), "t()")


# ---------------------------------------------------------------------------------------------
# rowsum() on tall double operands (>= 65536 rows: 16-bit group table + LDS accumulators; more groups than
# LDS holds: memory atomics), NA groups, NaN / NA / Inf values.  (Round 3 also built a row-panel form -- the
# group ids of a panel of rows staged in LDS for 16 columns at a time -- and measured it at 1.35 ms against
# 0.59 ms at BASELINE config 3: 41 nonzeros per wavefront between two barriers)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ngroup", [1, 7, 1000, 2300, 6000])
@pytest.mark.parametrize("shape", [(70_000, 37, 0.02), (200_001, 5, 0.3), (65_536, 130, 0.001), (100_003, 300, 0.05)])
def test_rowsum_tall_operands(hip, oracle, ngroup, shape):
    nrow, ncol, dens = shape
    cp, ri, v = random_csc(nrow, ncol, dens, seed=61)
    v = v.copy()
    rng = np.random.default_rng(62)
    if len(v) > 20:
        v[rng.choice(len(v), 6, replace=False)] = [np.nan, NA_real, np.inf, -np.inf, np.nan, 1e300]
    x = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, v)
    g = rng.integers(0, ngroup, nrow)
    grp = [None if rng.random() < 1e-4 else int(k) for k in g]          # a few NA groups
    for na_rm in (False, True):
        a, ua = hip.rowsum(x, grp, na_rm=na_rm)
        b, ub = oracle.rowsum(x, grp, na_rm=na_rm)
        assert ua == ub
        assert_equal(a, b, tol=1e-9, atol=1e-12, what=f"rowsum ngroup={ngroup} na_rm={na_rm}")
    # the prepared form: the 16-bit group id of every nonzero computed once (svt_dev_rowsum_prepare), then
    # products that stream values + ids (svt_dev_rowsum_prepared) -- same rules, src/rowsum_methods.c:44-64
    import torch
    from sparsearray_amd.device import DeviceCSC, RowsumPlan
    A = DeviceCSC.from_host(nrow, cp, ri, v)
    ug = hip._compute_ugroup(grp, nrow, True)
    gpos = hip._match(grp, ug)                         # 1-based positions, the NA group last (as R's match())
    for na_as_int in (False, True):                    # ... and as NA_integer_ -> last group (src/rowsum_methods.c:51-54)
        g32 = gpos.copy()
        if na_as_int and ug[-1] is None:
            g32[g32 == len(ug)] = -2147483648
        plan = RowsumPlan(A, torch.as_tensor(g32, device="cuda"), len(ug))
        for na_rm in (False, True):
            got = plan.run(na_rm=na_rm)
            torch.cuda.synchronize()
            b, _ = oracle.rowsum(x, grp, na_rm=na_rm)
            assert_equal(got.cpu().numpy().T, b, tol=1e-9, atol=1e-12,
                         what=f"prepared rowsum ngroup={ngroup} na_rm={na_rm} NA as integer: {na_as_int}")


@pytest.mark.gpu
@pytest.mark.parametrize("nnz", [1, 2, 3])
def test_prepared_rowsum_tiny_operands(hip, nnz):
    """RowsumPlan on operands with 1, 2, 3 nonzeros (the id kernel was launched with a grid of 0 blocks
    for exactly one nonzero); compute_rowsum_doubles, src/rowsum_methods.c:44-64."""
    import torch
    from sparsearray_amd.device import DeviceCSC, RowsumPlan
    nrow, ncol, ngroup = 9, 4, 3
    ri = np.array([5, 2, 8][:nnz], dtype=np.int32)
    v = np.array([1.5, -2.0, 4.0][:nnz])
    cp = np.array([0, 0, 1, min(nnz, 2), nnz], dtype=np.int64)       # leaf 0 empty; leaves 1, 2, 3 hold up to one nonzero each
    g32 = (np.arange(nrow, dtype=np.int32) % ngroup) + 1
    A = DeviceCSC.from_host(nrow, cp, ri, v)
    got = RowsumPlan(A, torch.as_tensor(g32, device="cuda"), ngroup).run()
    torch.cuda.synchronize()
    want = np.zeros((ncol, ngroup))
    for j in range(ncol):
        for k in range(cp[j], cp[j + 1]):
            want[j, g32[ri[k]] - 1] += v[k]
    assert np.array_equal(got.cpu().numpy(), want)


# ---------------------------------------------------------------------------------------------
# the bucketed transposition: paths the random shapes of test_hip_device_level.py do not reach
# ---------------------------------------------------------------------------------------------
def _check_transpose(m, dtype="double"):
    import torch
    from sparsearray_amd.device import DeviceCSC
    x = SVT_SparseArray.from_dense(m, type=dtype, lacunar=False)
    cp, ri, v = x.to_csc()
    tcp, tri, tv = x.t().to_csc()
    T = DeviceCSC.from_host(m.shape[0], cp, ri, v).t()
    torch.cuda.synchronize()
    assert np.array_equal(T.col_ptr.cpu().numpy(), tcp)
    assert np.array_equal(T.row_idx.cpu().numpy(), tri)
    assert np.array_equal(T.val.cpu().numpy(), tv)


def test_transpose_overfull_buckets(hip):
    """A band of 64 full rows: one fine bucket holds 32768 nonzeros (more than pass 3 ranks in one round) and
    one (group, coarse bucket) pair 16384 (more than pass 2 assembles in LDS): the round-by-round paths."""
    rng = np.random.default_rng(71)
    m = np.where(rng.random((4096, 512)) < 0.01, rng.normal(size=(4096, 512)), 0.0)
    m[128:192, :] = rng.normal(size=(64, 512))
    m[m == 0] = 0.0
    _check_transpose(m)
    _check_transpose(np.asfortranarray(m.T))            # wide: 512 rows of ~80 nonzeros, 4096 columns


def test_transpose_dense_and_odd_shapes(hip):
    rng = np.random.default_rng(72)
    _check_transpose(rng.normal(size=(600, 700)) * (rng.random((600, 700)) < 0.9))      # ~630 nonzeros per row: F = 8
    _check_transpose(rng.normal(size=(65, 257)) * (rng.random((65, 257)) < 0.3))        # one fine bucket + one row
    _check_transpose(rng.normal(size=(1, 300)))                                          # a single row
    _check_transpose(rng.normal(size=(300, 1)))                                          # a single column
    mi = (rng.integers(-5, 6, (3000, 40)) * (rng.random((3000, 40)) < 0.2)).astype(np.int32)
    _check_transpose(mi, "integer")


@pytest.mark.parametrize("spare", [32, 64, 128])
def test_product_with_cus_left_idle(hip, oracle, spare):
    """svt_dev_pbc_set_spare_cus: fewer row splits, every split's workgroups dealt round the XCDs -- the
    same product (clean and with a non-finite dense operand); the setting may change between the
    workspace query and the launch."""
    from test_hip_device_level import _dev
    from sparsearray_amd.device import PbcPlan, set_spare_cus, spare_cus
    nrow, ncol, K = 60000, 1300, 128                       # 3 column blocks x 2 dense tiles, 469 panels
    cp, ri, v = random_csc(nrow, ncol, 0.01, seed=71)
    x = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, v)
    rng = np.random.default_rng(72)
    A = _dev(cp, ri, v, nrow)
    plan = PbcPlan(A, K, 40, 16, 7)                        # sized with no CUs spared
    try:
        for poison in (False, True):
            y = rng.uniform(-1, 1, (nrow, K))
            if poison:
                y[nrow // 3, 5] = np.inf
            want = oracle.crossprod(x, y)
            Yd = torch.as_tensor(np.ascontiguousarray(y.T), device="cuda")
            outs = []
            for n in (0, spare):
                set_spare_cus(n)
                assert spare_cus() == n
                out = torch.full((K, ncol), 7.0, dtype=torch.float64, device="cuda")
                plan.run(Yd, nrow, out)
                torch.cuda.synchronize()
                assert_equal(out.cpu().numpy().T, want, tol=1e-9, atol=1e-11, what=f"spare={n} poison={poison}")
                outs.append(out)
            assert torch.allclose(outs[0], outs[1], rtol=1e-12, atol=1e-13, equal_nan=True)
        plan2 = PbcPlan(A, K, 40, 16, 7)                   # sized with CUs spared, run without
        set_spare_cus(0)
        out = torch.empty((K, ncol), dtype=torch.float64, device="cuda")
        plan2.run(Yd, nrow, out)
        torch.cuda.synchronize()
        assert_equal(out.cpu().numpy().T, want, tol=1e-9, atol=1e-11, what="sized spared, run unspared")
    finally:
        set_spare_cus(0)


# ---------------------------------------------------------------------------------------------
# sparse x sparse product by row panels (kernels_spmm.hip)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(50_000, 700, 40, 0.01, 0.02), (9_000, 300, 130, 0.05, 0.01), (100, 60, 7, 0.2, 0.05),
                                   (20_000, 50, 3, 0.3, 0.04), (70_001, 1200, 17, 0.004, 0.03)])
@pytest.mark.parametrize("dtype", ["double", "integer"])
def test_sparse_x_sparse_by_row_panels(hip, oracle, shape, dtype):
    """svt_dev_matmul_csc_csc against the oracle's x %*% y (the reference's C_crossprod2_SVT_SVT on t(x)) and
    against the dense route of the library."""
    from test_hip_device_level import _dev
    from sparsearray_amd.device import matmul_csc_csc
    nrow, ninner, K, da, db = shape
    cpa, ria, va = random_csc(nrow, ninner, da, seed=81)
    cpb, rib, vb = random_csc(ninner, K, db, seed=82)
    if dtype == "integer":
        va = np.round(va * 1000).astype(np.int32); vb = np.round(vb * 1000).astype(np.int32)
        va[va == 0] = 7; vb[vb == 0] = -3
    x = SVT_SparseArray.from_csc((nrow, ninner), dtype, cpa, ria, va)
    y = SVT_SparseArray.from_csc((ninner, K), dtype, cpb, rib, vb)
    want = oracle.matmul(x, y)
    A = _dev(cpa, ria, va, nrow)
    B = _dev(cpb, rib, vb, ninner)
    out, flag = matmul_csc_csc(A, B)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    got = out.cpu().numpy().T
    if dtype == "integer":
        assert_identical(got, want, what="sparse x sparse, integer")
    else:
        assert_equal(got, want, tol=1e-12, atol=1e-13, what="sparse x sparse")
    # the host-level entry point takes the same kernel for such operands
    assert_equal(hip.matmul(x, y), want, tol=1e-12, atol=1e-13, what="x %*% y")


def test_sparse_x_sparse_integer_sums_close_to_2_53(hip, oracle):
    """Integer operands whose cell sums come within a factor of four of 2^53: every product (< 2^49) and every
    partial sum is still an exact double, so the order in which the lane groups add (kernels_spmm.hip) cannot show --
    bit for bit with the oracle's ascending order (the reference adds integer products in double,
    src/SparseVec_dotprod.c:73-92).  Past 2^53 the reference itself rounds in its own order; not tested, stated."""
    from test_hip_device_level import _dev
    from sparsearray_amd.device import matmul_csc_csc
    nrow, ninner, K = 6000, 160, 9
    cpa, ria, va = random_csc(nrow, ninner, 0.5, seed=87)
    cpb, rib, vb = random_csc(ninner, K, 0.1, seed=88)
    rng = np.random.default_rng(89)
    lo, hi = int(2 ** 23.5), int(2 ** 24.5)
    va = (rng.integers(lo, hi, len(va)) * rng.choice([-1, 1], len(va))).astype(np.int32)
    vb = rng.integers(lo, hi, len(vb)).astype(np.int32)            # one sign: the sums do not cancel
    x = SVT_SparseArray.from_csc((nrow, ninner), "integer", cpa, ria, va)
    y = SVT_SparseArray.from_csc((ninner, K), "integer", cpb, rib, vb)
    want = oracle.matmul(x, y)
    top = np.abs(np.asarray(want)).max()
    # the bound that makes the order irrelevant, and "close": sums of |products| stay below 2^53, the largest cell above 2^51
    pos = np.abs(va.astype(np.float64))
    xa = SVT_SparseArray.from_csc((nrow, ninner), "double", cpa, ria, pos)
    ya = SVT_SparseArray.from_csc((ninner, K), "double", cpb, rib, vb.astype(np.float64))
    assert np.asarray(oracle.matmul(xa, ya)).max() < 2.0 ** 53
    assert top > 2.0 ** 51
    out, flag = matmul_csc_csc(_dev(cpa, ria, va, nrow), _dev(cpb, rib, vb, ninner))
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    assert_identical(out.cpu().numpy().T, want, what="integer sums close to 2^53")


def test_sparse_x_sparse_not_finite_takes_the_dense_route(hip, oracle):
    """A non-finite value or an NA in either operand: the flag goes up, and the entry point returns the
    reference's result (its dirty-leaf loops multiply the implicit zeros too)."""
    from test_hip_device_level import _dev
    from sparsearray_amd.device import matmul_csc_csc
    nrow, ninner, K = 3000, 90, 11
    cpa, ria, va = random_csc(nrow, ninner, 0.05, seed=83)
    cpb, rib, vb = random_csc(ninner, K, 0.04, seed=84)
    for which, poison in (("x", np.inf), ("y", np.nan), ("x", NA_real)):
        va2, vb2 = va.copy(), vb.copy()
        (va2 if which == "x" else vb2)[5] = poison
        x = SVT_SparseArray.from_csc((nrow, ninner), "double", cpa, ria, va2)
        y = SVT_SparseArray.from_csc((ninner, K), "double", cpb, rib, vb2)
        _, flag = matmul_csc_csc(_dev(cpa, ria, va2, nrow), _dev(cpb, rib, vb2, ninner))
        torch.cuda.synchronize()
        assert int(flag.item()) == 1
        assert_equal(hip.matmul(x, y), oracle.matmul(x, y), tol=1e-12, atol=1e-13, strict_na=True, what=f"{which} {poison}")


def test_sparse_x_sparse_random_shapes(hip):
    """Edge shapes of the row-panel kernel (one panel, a ragged last panel, empty columns and rows, K = 1,
    more columns than a workgroup takes, dense-ish operands) against a dense numpy product."""
    from test_hip_device_level import _dev
    from sparsearray_amd.device import matmul_csc_csc
    rng = np.random.default_rng(91)
    shapes = [(1, 1, 1), (63, 5, 1), (64, 1, 3), (129, 40, 19), (8191, 9, 2), (8193, 33, 40), (16385, 3, 70),
              (30000, 200, 5), (5000, 64, 300), (70000, 17, 33)]
    for nrow, ninner, K in shapes:
        for da, db in ((0.3, 0.5), (0.02, 0.05), (0.0, 0.1), (0.1, 0.0)):
            a = (rng.random((nrow, ninner)) < da) * rng.uniform(-2, 2, (nrow, ninner))
            b = (rng.random((ninner, K)) < db) * rng.uniform(-2, 2, (ninner, K))
            if ninner > 2:
                a[:, ninner // 2] = 0.0                      # an empty column of A
                b[ninner // 3, :] = 0.0                      # an empty row of B
            def csc(m):
                cp = np.zeros(m.shape[1] + 1, dtype=np.int64)
                ri, vv = [], []
                for j in range(m.shape[1]):
                    nz = np.nonzero(m[:, j])[0]
                    ri.append(nz); vv.append(m[nz, j]); cp[j + 1] = cp[j] + len(nz)
                return cp, (np.concatenate(ri) if ri else np.zeros(0)).astype(np.int32), \
                    (np.concatenate(vv) if vv else np.zeros(0)).astype(np.float64)
            A = _dev(*csc(a), nrow)
            B = _dev(*csc(b), ninner)
            out, flag = matmul_csc_csc(A, B)
            torch.cuda.synchronize()
            assert int(flag.item()) == 0
            want = a @ b
            got = out.cpu().numpy().T
            assert got.shape == want.shape
            assert np.allclose(got, want, rtol=1e-12, atol=1e-12), (nrow, ninner, K, da, db, np.abs(got - want).max())


def test_sparse_x_sparse_prepared_operand(hip, oracle):
    """SpmmPlan: the table and the value scan of A once, several B's afterwards -- same results as the one-call
    form; a non-finite value in A is remembered by the plan, one in B is seen per product."""
    from test_hip_device_level import _dev
    from sparsearray_amd.device import SpmmPlan, matmul_csc_csc
    nrow, ninner = 40_000, 500
    cpa, ria, va = random_csc(nrow, ninner, 0.02, seed=85)
    A = _dev(cpa, ria, va, nrow)
    plan = SpmmPlan(A)
    for K, seed in ((9, 86), (40, 87)):
        cpb, rib, vb = random_csc(ninner, K, 0.03, seed=seed)
        B = _dev(cpb, rib, vb, ninner)
        out, flag = plan.run(B)
        ref, _ = matmul_csc_csc(A, B)
        torch.cuda.synchronize()
        assert int(flag.item()) == 0
        assert torch.allclose(out, ref, rtol=1e-13, atol=1e-13)
        x = SVT_SparseArray.from_csc((nrow, ninner), "double", cpa, ria, va)
        y = SVT_SparseArray.from_csc((ninner, K), "double", cpb, rib, vb)
        assert_equal(out.cpu().numpy().T, oracle.matmul(x, y), tol=1e-12, atol=1e-13, what="prepared")
        vb2 = vb.copy(); vb2[0] = np.inf
        _, flag = plan.run(_dev(cpb, rib, vb2, ninner))
        torch.cuda.synchronize()
        assert int(flag.item()) == 1                      # B's Inf
        _, flag = plan.run(B)
        torch.cuda.synchronize()
        assert int(flag.item()) == 0                      # ... does not stick
    va2 = va.copy(); va2[len(va2) // 2] = np.nan
    plan2 = SpmmPlan(_dev(cpa, ria, va2, nrow))
    _, flag = plan2.run(B)
    torch.cuda.synchronize()
    assert int(flag.item()) == 1


@pytest.mark.parametrize("shape", [(70_000, 300, 0.01, 1), (20_000, 64, 0.05, 1), (3_000, 40, 0.1, 1), (40_000, 120, 0.02, 4)])
def test_rowsums_with_prepared_table(hip, shape):
    """RowSumsPlan (svt_dev_rowsums_prepare / _prepared): the one-call rowSums, na.rm both ways (the LDS additions of
    a row come in the order the lane groups get to them: equal up to the last bits)."""
    from test_hip_device_level import _dev
    from sparsearray_amd.device import RowSumsPlan, rowsums
    nrow, ncol, dens, inner = shape
    cp, ri, v = random_csc(nrow, ncol, dens, seed=95)
    v = v.copy(); v[3] = np.nan
    A = _dev(cp, ri, v, nrow)
    plan = RowSumsPlan(A, inner)
    for na_rm in (False, True):
        a = plan.run(na_rm=na_rm)
        b = rowsums(A, na_rm=na_rm, inner=inner)
        torch.cuda.synchronize()
        assert torch.allclose(a, b, rtol=1e-12, atol=1e-13, equal_nan=True)


@pytest.mark.parametrize("nrow", [4_700_000, 4_600_000])
def test_layout_build_count_pass_both_forms(hip, oracle, nrow):
    """The LDS-DMA layout's count pass: one stream over the offsets with the panels' counters in LDS (up to
    36000 panels of 128 rows), the chunked walk beyond -- the same product either way."""
    from test_hip_device_level import _dev
    from sparsearray_amd.device import PbcPlan
    ncol, K = 90, 8
    cp, ri, v = random_csc(nrow, ncol, 0.0008, seed=97)
    x = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, v)
    x.svt_is_null = False
    rng = np.random.default_rng(98)
    y = rng.uniform(-1, 1, (nrow, K))
    want = oracle.crossprod(x, y)
    A = _dev(cp, ri, v, nrow)
    plan = PbcPlan(A, K, 40, 16, 7)
    Yd = torch.as_tensor(np.ascontiguousarray(y.T), device="cuda")
    out = torch.full((K, ncol), 7.0, dtype=torch.float64, device="cuda")
    plan.run(Yd, nrow, out)
    torch.cuda.synchronize()
    assert_equal(out.cpu().numpy().T, want, tol=1e-9, atol=1e-11, what=f"nrow={nrow}")


def test_layout_pool_is_bounded_and_trims(hip):
    """Layout buffers come from the library's own stream-ordered pool: a released layout stays cached for the next
    build (up to 3 GiB), svt_dev_pbc_trim() hands the cache back to the driver, and a plan can be dropped while a
    product on another stream is still running (release behind events, no device-wide synchronisation)."""
    import torch
    from test_hip_device_level import _dev
    from sparsearray_amd.device import PbcPlan, trim_layout_pool
    cp, ri, v = random_csc(300_000, 2000, 0.01, seed=77)          # 6e6 nonzeros: ~80 MB of records
    A = _dev(cp, ri, v, 300_000)
    K = 64
    Yd = torch.as_tensor(np.random.default_rng(78).uniform(-1, 1, (K, 300_000)), device="cuda")
    out = torch.empty((K, 2000), dtype=torch.float64, device="cuda")
    ref = None
    side = torch.cuda.Stream()
    for _ in range(3):
        plan = PbcPlan(A, K)
        with torch.cuda.stream(side):
            plan.run(Yd, 300_000, out)
        del plan                                                  # released while the product may still be in flight
        side.synchronize()
        if ref is None:
            ref = out.clone()
        assert torch.equal(out, ref)
    torch.cuda.synchronize()
    free_cached, _ = torch.cuda.mem_get_info()
    trim_layout_pool()
    torch.cuda.synchronize()
    free_trimmed, _ = torch.cuda.mem_get_info()
    assert free_trimmed >= free_cached + 50 * 2 ** 20             # the cached layout (~80 MB) went back


def test_rowsums_whole_column_pipeline_long_leaves(hip, oracle):
    """rowSums(x, dims = 2) / rowCountNAs through the persistent whole-column kernel (rowstats_whole_pipe_kernel: the next
    column's leaves are fetched while the current column's cells leave): leaves longer than the two trips it fetches
    ahead (300 and 1000 nonzeros), an empty leaf, NaN / NA values, na.rm -- src/SparseArray_matrixStats.c:774-913."""
    shape = (9000, 1100, 6)
    rng = np.random.default_rng(131)
    a = np.where(rng.random(shape) < 0.01, rng.normal(size=shape), 0.0)
    a[rng.choice(shape[0], 300, replace=False), 5, 2] = rng.normal(size=300)
    a[rng.choice(shape[0], 1000, replace=False), 700, 0] = rng.normal(size=1000)
    a[:, 9, 4] = 0.0
    a[17, 5, 2] = np.nan
    a[18, 700, 0] = NA_real
    x = SVT_SparseArray.from_dense(np.asfortranarray(a), "double", lacunar=False)
    for na_rm in (False, True):
        assert_equal(hip.rowSums(x, dims=2, na_rm=na_rm), oracle.rowSums(x, dims=2, na_rm=na_rm), tol=1e-9, atol=1e-12,
                     what=f"rowSums dims=2 na_rm={na_rm}")
    assert_identical(hip.rowCountNAs(x, dims=2), oracle.rowCountNAs(x, dims=2), "rowCountNAs dims=2")


@pytest.mark.parametrize("dtype", ["double", "integer"])
def test_rowsums_of_many_short_leaves_per_output_column(hip, oracle, dtype):
    """rowSums / rowCountNAs of a 4-d array over its last 3, 2 and 1 axes: 144000 / 9000 leaves per output column go through
    the persistent whole-column kernel in chunks of 64 leaves whose cells are added to `out` (rowstats_whole_pipe_kernel
    <T, true>), 60 leaves per column through its plain form -- against the oracle (src/SparseArray_matrixStats.c:774-913);
    integer input bit for bit."""
    shape = (8500, 16, 150, 60)
    ncol = int(np.prod(shape[1:]))
    cp, ri, v = random_csc(shape[0], ncol, 0.01, seed=141, dtype="double" if dtype == "double" else "int")
    v = v.copy()
    rng = np.random.default_rng(142)
    hit = rng.choice(len(v), 12, replace=False)
    if dtype == "double":
        v[hit] = [np.nan, NA_real, np.inf, -np.inf, 1e300, -1e300, np.nan, NA_real, 3.5, -2.0, 0.25, 7.0]
    else:
        v[hit[:4]] = NA_integer
    x = SVT_SparseArray.from_csc((shape[0], ncol), dtype, cp, ri, v)
    x = SVT_SparseArray(shape, x.type, x.leaves)
    for dims in (1, 2, 3):
        for na_rm in (False, True):
            got, want = hip.rowSums(x, dims=dims, na_rm=na_rm), oracle.rowSums(x, dims=dims, na_rm=na_rm)
            if dtype == "integer":
                assert_identical(got, want, f"rowSums dims={dims} na_rm={na_rm}")
            else:
                assert_equal(got, want, tol=1e-9, atol=1e-9, what=f"rowSums dims={dims} na_rm={na_rm}")
        assert_identical(hip.rowCountNAs(x, dims=dims), oracle.rowCountNAs(x, dims=dims), f"rowCountNAs dims={dims}")
