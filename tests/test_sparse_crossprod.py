"""crossprod(x) / crossprod(x, y) of two sparse operands without a dense buffer (kernels_gram.hip,
svt_dev_crossprod_csc_csc) against the CPU oracle's C_crossprod1_SVT / C_crossprod2_SVT_SVT
(src/SparseMatrix_mult.c:1037-1140) and against the library's own dense-buffer route."""
import numpy as np
import pytest
import torch

from helpers import assert_equal, assert_identical, random_csc
from sparsearray_amd import NA_integer, NA_real, SVT_SparseArray

pytestmark = pytest.mark.gpu


def _dev(cp, ri, v, nrow):
    from sparsearray_amd.device import DeviceCSC
    return DeviceCSC.from_host(nrow, cp, ri, v)


def _ints(v, seed):
    w = np.round(v * 1000).astype(np.int32)
    w[w == 0] = 7
    return w


# (nrow, ncol, density): short and tall operands, rows with none / one / many nonzeros
SHAPES = [(3000, 700, 0.05), (25000, 400, 0.07), (100, 60, 0.2), (50_000, 1300, 0.004), (64, 1, 0.5), (1, 9, 0.7),
          (7000, 257, 0.3)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dtype", ["double", "integer"])
def test_unary_crossprod_device_level(oracle, shape, dtype):
    """Symmetric form: cells c <= j in LDS, columns j and n - 1 - j per workgroup, mirrored."""
    from sparsearray_amd.device import crossprod_csc_csc
    nrow, ncol, d = shape
    cp, ri, v = random_csc(nrow, ncol, d, seed=601)
    if dtype == "integer":
        v = _ints(v, 1)
    x = SVT_SparseArray.from_csc((nrow, ncol), dtype, cp, ri, v)
    want = np.asarray(oracle.crossprod(x))
    A = _dev(cp, ri, v, nrow)
    out, flag = crossprod_csc_csc(A.t(), A, sym=True)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    got = out.cpu().numpy().T
    assert np.array_equal(got, got.T), "crossprod(x) must be bit-symmetric"
    if dtype == "integer":
        assert_identical(got, want, what="crossprod(x), integer")
    else:
        assert_equal(got, want, tol=1e-12, atol=1e-13, what="crossprod(x)")
    # the general form on the same operands (no symmetry used) gives the same cells
    out2, flag2 = crossprod_csc_csc(A.t(), A, sym=False)
    torch.cuda.synchronize()
    assert int(flag2.item()) == 0
    assert_equal(out2.cpu().numpy().T, want, tol=1e-12, atol=1e-13, what="crossprod(x, x)")


@pytest.mark.parametrize("shape", [(3000, 700, 90, 0.05, 0.02), (25000, 400, 650, 0.07, 0.2), (100, 60, 7, 0.2, 0.05),
                                   (20_000, 50, 3, 0.3, 0.04), (70_001, 1200, 17, 0.004, 0.03), (500, 1, 1, 0.5, 0.5)])
@pytest.mark.parametrize("types", [("double", "double"), ("integer", "integer")])
def test_binary_crossprod_device_level(hip, oracle, shape, types):
    from sparsearray_amd.device import crossprod_csc_csc
    nrow, nx, ny, dx, dy = shape
    cpx, rix, vx = random_csc(nrow, nx, dx, seed=611)
    cpy, riy, vy = random_csc(nrow, ny, dy, seed=612)
    if types[0] == "integer":
        vx, vy = _ints(vx, 1), _ints(vy, 2)
    x = SVT_SparseArray.from_csc((nrow, nx), types[0], cpx, rix, vx)
    y = SVT_SparseArray.from_csc((nrow, ny), types[1], cpy, riy, vy)
    want = np.asarray(oracle.crossprod(x, y))
    X, Y = _dev(cpx, rix, vx, nrow), _dev(cpy, riy, vy, nrow)
    out, flag = crossprod_csc_csc(X.t(), Y)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    got = out.cpu().numpy().T
    if types[0] == "integer":
        assert_identical(got, want, what="crossprod(x, y), integer")
    else:
        assert_equal(got, want, tol=1e-12, atol=1e-13, what="crossprod(x, y)")
    # host entry points: whichever route they choose, the reference's result
    assert_equal(hip.crossprod(x, y), want, tol=1e-12, atol=1e-13, what="host crossprod(x, y)")
    assert_equal(hip.crossprod(y, x), want.T, tol=1e-12, atol=1e-13, what="host crossprod(y, x)")


@pytest.mark.parametrize("panel", [(0, 6), (0, 9), (100, 8), (0, 13), (300, 14)])
def test_wide_results_go_by_cell_panels(oracle, panel):
    """Results taller than one workgroup's LDS: panels of cells + the table of run bounds (forced here on small
    operands by shrinking the panel; the default is one block up to 16384 / 20400 cells, panels of 8192 beyond)."""
    from sparsearray_amd.device import crossprod_csc_csc, set_sparse_crossprod_panel
    nrow, nx, ny = 4000, 777, 333
    cpx, rix, vx = random_csc(nrow, nx, 0.03, seed=621)
    cpy, riy, vy = random_csc(nrow, ny, 0.05, seed=622)
    x = SVT_SparseArray.from_csc((nrow, nx), "double", cpx, rix, vx)
    y = SVT_SparseArray.from_csc((nrow, ny), "double", cpy, riy, vy)
    X, Y = _dev(cpx, rix, vx, nrow), _dev(cpy, riy, vy, nrow)
    try:
        set_sparse_crossprod_panel(*panel)
        out, flag = crossprod_csc_csc(X.t(), Y)
        outs, flags = crossprod_csc_csc(X.t(), X, sym=True)
        torch.cuda.synchronize()
    finally:
        set_sparse_crossprod_panel(-1, -1)
    assert int(flag.item()) == 0 and int(flags.item()) == 0
    assert_equal(out.cpu().numpy().T, np.asarray(oracle.crossprod(x, y)), tol=1e-12, atol=1e-13, what="panels")
    gs = outs.cpu().numpy().T
    assert np.array_equal(gs, gs.T)
    assert_equal(gs, np.asarray(oracle.crossprod(x)), tol=1e-12, atol=1e-13, what="panels, symmetric")


def test_really_wide_result(oracle):
    """12 000 result cells per column: past what two workgroups per CU hold (10 200), one block of 96 KB of LDS, one
    workgroup per CU (the default up to 16 384 columns for the symmetric form; the cell-panel form beyond is forced on
    small operands by test_wide_results_go_by_cell_panels and the fuzzer, and runs at its default on 17 000 columns
    below); against scipy."""
    import scipy.sparse as sp
    from sparsearray_amd.device import crossprod_csc_csc
    nrow, ncol = 3000, 12_000
    cp, ri, v = random_csc(nrow, ncol, 0.01, seed=631)
    A = _dev(cp, ri, v, nrow)
    out, flag = crossprod_csc_csc(A.t(), A, sym=True)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    got = out.cpu().numpy()
    m = sp.csc_matrix((v, ri, cp), shape=(nrow, ncol))
    want = (m.T @ m).toarray()
    assert np.array_equal(got, got.T)
    assert np.allclose(got, want, rtol=1e-12, atol=1e-12)
    # 17 000 columns: the symmetric form goes by panels of 8192 cells at its defaults, the general form in one block
    nrow, ncol = 2500, 17_000
    cp, ri, v = random_csc(nrow, ncol, 0.004, seed=632)
    A = _dev(cp, ri, v, nrow)
    At = A.t()
    out, flag = crossprod_csc_csc(At, A, sym=True)
    out2, flag2 = crossprod_csc_csc(At, A, sym=False)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0 and int(flag2.item()) == 0
    m = sp.csc_matrix((v, ri, cp), shape=(nrow, ncol))
    want = (m.T @ m).toarray()
    got = out.cpu().numpy()
    assert np.array_equal(got, got.T)
    assert np.allclose(got, want, rtol=1e-12, atol=1e-12)
    assert np.allclose(out2.cpu().numpy(), want, rtol=1e-12, atol=1e-12)


def test_not_finite_raises_the_flag_and_the_entry_points_follow_the_reference(hip, oracle):
    """A non-finite value or an NA anywhere: the flag goes up; svt_crossprod1_SVT / svt_crossprod2_SVT_SVT then
    answer through the dense-buffer route, whose dirty-leaf rules are the reference's
    (src/SparseMatrix_mult.c:632-724, 827-873)."""
    from sparsearray_amd.device import crossprod_csc_csc
    nrow, nx, ny = 3000, 90, 40
    cpx, rix, vx = random_csc(nrow, nx, 0.05, seed=641)
    cpy, riy, vy = random_csc(nrow, ny, 0.04, seed=642)
    for which, poison in (("x", np.inf), ("y", np.nan), ("x", NA_real), ("y", -np.inf)):
        vx2, vy2 = vx.copy(), vy.copy()
        (vx2 if which == "x" else vy2)[5] = poison
        x = SVT_SparseArray.from_csc((nrow, nx), "double", cpx, rix, vx2)
        y = SVT_SparseArray.from_csc((nrow, ny), "double", cpy, riy, vy2)
        X, Y = _dev(cpx, rix, vx2, nrow), _dev(cpy, riy, vy2, nrow)
        _, flag = crossprod_csc_csc(X.t(), Y)
        torch.cuda.synchronize()
        assert int(flag.item()) == 1, (which, poison)
        assert_equal(hip.crossprod(x, y), oracle.crossprod(x, y), tol=1e-12, atol=1e-13, strict_na=True,
                     what=f"{which} {poison}")
        op = x if which == "x" else y
        O = X if which == "x" else Y
        _, flag = crossprod_csc_csc(O.t(), O, sym=True)
        torch.cuda.synchronize()
        assert int(flag.item()) == 1
        assert_equal(hip.crossprod(op), oracle.crossprod(op), tol=1e-12, atol=1e-13, strict_na=True,
                     what=f"crossprod({which}) {poison}")
    # integer NA
    vxi = _ints(vx, 1)
    vxi[11] = NA_integer
    xi = SVT_SparseArray.from_csc((nrow, nx), "integer", cpx, rix, vxi)
    Xi = _dev(cpx, rix, vxi, nrow)
    _, flag = crossprod_csc_csc(Xi.t(), Xi, sym=True)
    torch.cuda.synchronize()
    assert int(flag.item()) == 1
    assert_equal(hip.crossprod(xi), oracle.crossprod(xi), tol=0, strict_na=True, what="integer NA")


@pytest.mark.parametrize("shape", [(25000, 400, 0.07), (25000, 650, 0.2), (60_000, 900, 0.01)])
@pytest.mark.parametrize("route", [1.0, 0.0, -1.0])
def test_host_entry_points_choose_a_route(hip, oracle, shape, route):
    """The reference's published benchmark shapes (inst/scripts/benchmark_crossprod.R:123-166: 25000 x 400 @ 0.07,
    25000 x 650 @ 0.20) and a sparser one through svt_crossprod1_SVT and svt_crossprod2_SVT_SVT: with the measured
    route model, with the sparse-aware kernel forced and with the dense-buffer route forced."""
    from sparsearray_amd.device import set_sparse_crossprod_cost
    nrow, ncol, d = shape
    cp, ri, v = random_csc(nrow, ncol, d, seed=651)
    x = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, v)
    cp2, ri2, v2 = random_csc(nrow, 37, 0.03, seed=652)
    y = SVT_SparseArray.from_csc((nrow, 37), "double", cp2, ri2, v2)
    want = np.asarray(oracle.crossprod(x))
    want2 = np.asarray(oracle.crossprod(x, y))
    try:
        set_sparse_crossprod_cost(route)
        got = np.asarray(hip.crossprod(x))
        got_xx = np.asarray(hip.crossprod(x, x))
        got2 = np.asarray(hip.crossprod(x, y))
        got3 = np.asarray(hip.crossprod(y, x))
    finally:
        set_sparse_crossprod_cost(1.0)
    assert np.array_equal(got, got.T)
    assert_equal(got, want, tol=1e-12, atol=1e-12, what="crossprod(x)")
    assert_equal(got_xx, want, tol=1e-12, atol=1e-12, what="crossprod(x, x)")
    assert_equal(got2, want2, tol=1e-12, atol=1e-12, what="crossprod(x, y)")
    assert_equal(got3, want2.T, tol=1e-12, atol=1e-12, what="crossprod(y, x)")


@pytest.mark.parametrize("lacunar", [True, False], ids=["lacunar", "plain"])
def test_golden_unary_and_sparse_sparse_products_through_the_sparse_kernel(hip, lacunar):
    """The reference's own test vectors (tests/testthat/test-SparseMatrix-mult.R:206-304 via tests/golden) for
    crossprod() / tcrossprod() with the sparse-aware kernel forced wherever both operands are sparse and finite
    (non-finite operands raise the flag and fall through to the dense-buffer route -- the NA / NaN / Inf cases of
    those vectors check exactly that)."""
    from helpers import check_case, golden_cases
    from sparsearray_amd.device import set_sparse_crossprod_cost
    n = 0
    try:
        set_sparse_crossprod_cost(0.0)
        for case in golden_cases():
            if case["fn"] in ("crossprod", "tcrossprod"):
                check_case(hip, case, lacunar=lacunar, gpu=True)
                n += 1
    finally:
        set_sparse_crossprod_cost(1.0)
    assert n >= 200


def test_rows_of_very_unequal_length(hip, oracle):
    """The route choice estimates the pairs of nonzeros that meet in a row as nnz^2 / (2 nrow); an operand with a few
    hundred nearly full rows holds far more of them.  With t(x) built the symmetric route counts them exactly and
    hands such an operand to the other route; whichever runs, the reference's cells."""
    from sparsearray_amd.device import set_sparse_crossprod_cost
    rng = np.random.default_rng(661)
    nrow, ncol = 40_000, 900
    a = (rng.random((nrow, ncol)) < 0.01) * rng.normal(size=(nrow, ncol))
    heavy = rng.choice(nrow, size=400, replace=False)
    a[heavy, :] = rng.normal(size=(400, ncol))
    x = SVT_SparseArray.from_dense(np.asfortranarray(a), "double", lacunar=False)
    want = np.asarray(oracle.crossprod(x))
    for route in (1.0, 0.0):
        try:
            set_sparse_crossprod_cost(route)
            got = np.asarray(hip.crossprod(x))
        finally:
            set_sparse_crossprod_cost(1.0)
        assert np.array_equal(got, got.T)
        assert_equal(got, want, tol=1e-11, atol=1e-11, what=f"route {route}")


@pytest.mark.parametrize("shape", [(400, 25_000, 0.07), (700, 3000, 0.05), (60, 100, 0.2), (1, 64, 0.5), (9, 1, 0.7)])
@pytest.mark.parametrize("route", [1.0, 0.0, -1.0])
def test_tcrossprod_in_one_call(hip, oracle, shape, route):
    """tcrossprod(x) / tcrossprod(x, y) through svt_tcrossprod1_SVT / svt_tcrossprod2_SVT_SVT (operands transposed on the
    device; the sparse-aware kernel walks x itself) against the oracle's reference flow t() + C_crossprod1_SVT /
    C_crossprod2_SVT_SVT (R/SparseMatrix-mult.R:165-193), each route forced and the model's choice; NA / Inf operands
    follow the reference's dirty-leaf rules."""
    from sparsearray_amd.device import set_sparse_crossprod_cost
    nrow, ncol, d = shape
    cp, ri, v = random_csc(nrow, ncol, d, seed=671)
    x = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, v)
    ny = max(1, nrow // 3)
    cp2, ri2, v2 = random_csc(ny, ncol, min(1.0, d * 2), seed=672)
    y = SVT_SparseArray.from_csc((ny, ncol), "double", cp2, ri2, v2)
    vbad = v.copy()
    if len(vbad) > 3:
        vbad[3] = np.inf
    xbad = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, vbad)
    try:
        set_sparse_crossprod_cost(route)
        got1, got2, got3 = hip.tcrossprod(x), hip.tcrossprod(x, y), hip.tcrossprod(y, x)
        gotb = hip.tcrossprod(xbad)
    finally:
        set_sparse_crossprod_cost(1.0)
    g1 = np.asarray(got1)
    assert np.array_equal(g1, g1.T)
    assert_equal(got1, oracle.tcrossprod(x), tol=1e-11, atol=1e-12, what="tcrossprod(x)")
    want2 = np.asarray(oracle.tcrossprod(x, y))
    assert_equal(got2, want2, tol=1e-11, atol=1e-12, what="tcrossprod(x, y)")
    assert_equal(got3, want2.T, tol=1e-11, atol=1e-12, what="tcrossprod(y, x)")
    assert_equal(gotb, oracle.tcrossprod(xbad), tol=1e-11, atol=1e-12, strict_na=True, what="tcrossprod(x with Inf)")
