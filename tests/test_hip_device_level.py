"""Device-level entry points (operands resident in HBM, torch owns the memory):
the panel-blocked crossprod fast path and the stats kernels against the CPU
oracle."""
import numpy as np
import pytest
import torch

from helpers import assert_equal, random_csc
from sparsearray_amd import NA_real, SVT_SparseArray

pytestmark = pytest.mark.gpu


def _dev(cp, ri, v, nrow):
    from sparsearray_amd.device import DeviceCSC
    return DeviceCSC.from_host(nrow, cp, ri, v)


@pytest.mark.parametrize("cfg", [(32, 16, 7), (16, 16, 7), (40, 16, 7), (24, 16, 7), (5, 16, 7), (48, 8, 7),
                                 (64, 8, 7), (32, 8, 6), (64, 4, 5), (32, 16, 8), (20, 16, 8)])
@pytest.mark.parametrize("shape", [(5000, 300, 70), (70000, 1100, 128), (300, 17, 5), (4096, 700, 64),
                                   (1281, 90, 3), (1282, 90, 130)])
def test_pbc_crossprod_matches_oracle(hip, oracle, cfg, shape):
    from sparsearray_amd.device import PbcPlan
    nrow, ncol, K = shape
    CBW, WPB, logR = cfg
    cp, ri, v = random_csc(nrow, ncol, 0.01 if nrow > 1000 else 0.2, seed=21)
    x = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, v)
    y = np.random.default_rng(22).uniform(-1, 1, (nrow, K))
    want = oracle.crossprod(x, y)
    A = _dev(cp, ri, v, nrow)
    Yd = torch.as_tensor(np.ascontiguousarray(y.T), device="cuda")      # (K, nrow) == col-major
    out = torch.zeros((K, ncol), dtype=torch.float64, device="cuda")
    plan = PbcPlan(A, K, CBW, WPB, logR)
    plan.run(Yd, nrow, out)
    torch.cuda.synchronize()
    assert_equal(out.cpu().numpy().T, want, tol=1e-9, atol=1e-11, what="pbc crossprod")
    # mat_SVT orientation: out[k, c] with strides (K, 1) and transposed dense operand
    out2 = torch.zeros((ncol, K), dtype=torch.float64, device="cuda")
    Yr = torch.as_tensor(np.ascontiguousarray(y), device="cuda")        # (nrow, K): K x nrow col-major
    plan.run(Yr, K, out2, stride_c=K, stride_k=1, tr_y=True)
    torch.cuda.synchronize()
    assert_equal(out2.cpu().numpy(), want, tol=1e-9, atol=1e-11, what="pbc crossprod tr")


def test_pbc_special_values_take_general_path(hip, oracle):
    from sparsearray_amd.device import PbcPlan
    nrow, ncol, K = 4000, 150, 20
    cp, ri, v = random_csc(nrow, ncol, 0.02, seed=23)
    v = v.copy()
    v[5] = NA_real
    v[100] = np.inf
    x = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, v)
    A = _dev(cp, ri, v, nrow)
    plan = PbcPlan(A, K)
    rng = np.random.default_rng(24)
    for poison in ([], [np.nan], [np.inf, NA_real]):
        y = rng.uniform(-1, 1, (nrow, K))
        for t, val in enumerate(poison):
            y[17 + 31 * t, 3 + t] = val
        want = oracle.crossprod(x, y)
        Yd = torch.as_tensor(np.ascontiguousarray(y.T), device="cuda")
        out = torch.zeros((K, ncol), dtype=torch.float64, device="cuda")
        plan.run(Yd, nrow, out)
        torch.cuda.synchronize()
        assert_equal(out.cpu().numpy().T, want, tol=1e-9, atol=1e-11, strict_na=True,
                     what=f"poison={poison}")


@pytest.mark.parametrize("cbw", [32, 40])
def test_pbc_dma_path_prescan_sees_every_dense_entry(hip, oracle, cbw):
    """The LDS-DMA kernel never sees Y in registers: its finiteness prescan reads
    each workgroup's share of the landed panels back from LDS.  One poisoned
    entry anywhere (first/last row, last partial panel, tail of K, every column
    block's share) must switch the whole product to the general semantics."""
    from sparsearray_amd.device import PbcPlan
    nrow, ncol, K = 20000 + 77, 2100, 70
    cp, ri, v = random_csc(nrow, ncol, 0.005, seed=31)
    x = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, v)
    A = _dev(cp, ri, v, nrow)
    plan = PbcPlan(A, K, cbw, 16, 7)
    rng = np.random.default_rng(32)
    y0 = rng.uniform(-1, 1, (nrow, K))
    spots = [(0, 0), (nrow - 1, K - 1), (nrow - 1, 0), (0, K - 1), (127, 63), (128, 64), (nrow - 70, 5),
             (12345, 69)] + [(int(r), int(k)) for r, k in zip(rng.integers(0, nrow, 6), rng.integers(0, K, 6))]
    out = torch.zeros((K, ncol), dtype=torch.float64, device="cuda")
    for r, k in spots:
        y = y0.copy()
        y[r, k] = np.inf if (r + k) % 2 else np.nan
        want = oracle.crossprod(x, y)
        Yd = torch.as_tensor(np.ascontiguousarray(y.T), device="cuda")
        plan.run(Yd, nrow, out)
        torch.cuda.synchronize()
        assert_equal(out.cpu().numpy().T, want, tol=1e-9, atol=1e-11, strict_na=True, what=f"poison at {(r, k)}")
    want = oracle.crossprod(x, y0)
    Yd = torch.as_tensor(np.ascontiguousarray(y0.T), device="cuda")
    plan.run(Yd, nrow, out)
    torch.cuda.synchronize()
    assert_equal(out.cpu().numpy().T, want, tol=1e-9, atol=1e-11, what="clean")


@pytest.mark.parametrize("case", ["dense", "one_column", "wide_K", "min_rows", "below_min_rows", "two_panels",
                                  "two_panels_ragged", "three_panels", "empty", "one_nonzero", "tall_thin"])
def test_pbc_odd_shapes(hip, oracle, case):
    """Shapes at the edges of the panel kernels' preconditions."""
    from sparsearray_amd.device import PbcPlan
    nrow, ncol, K, dens = {"dense": (700, 90, 64, 0.5), "one_column": (5000, 1, 64, 0.3),
                           "wide_K": (3000, 200, 200, 0.02), "min_rows": (192, 50, 10, 0.1),
                           "below_min_rows": (191, 50, 10, 0.1), "two_panels": (193, 50, 10, 0.1),
                           "two_panels_ragged": (255, 50, 70, 0.1), "three_panels": (288, 50, 64, 0.1),
                           "empty": (1000, 70, 8, 0.0),
                           "one_nonzero": (1000, 70, 8, 0.0), "tall_thin": (200000, 3, 5, 0.001)}[case]
    cp, ri, v = random_csc(nrow, ncol, dens, seed=51)
    if case == "one_nonzero":
        cp = np.zeros(ncol + 1, dtype=np.int64); cp[41:] = 1
        ri = np.array([999], dtype=np.int32); v = np.array([-2.5])
    x = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, v)
    # the device layout has leaves only: an all-empty operand is a tree of NULL
    # leaves, not the `SVT == NULL` object the reference short-circuits on
    # (src/SparseMatrix_mult.c:389-390); that flag is host-level business
    x.svt_is_null = False
    rng = np.random.default_rng(52)
    A = _dev(cp, ri, v, nrow)
    plan = PbcPlan(A, K)
    for poison in (False, True):
        y = rng.uniform(-1, 1, (nrow, K))
        if poison:
            y[nrow // 2, K - 1] = np.nan
        want = oracle.crossprod(x, y)
        Yd = torch.as_tensor(np.ascontiguousarray(y.T), device="cuda")
        out = torch.full((K, ncol), 7.0, dtype=torch.float64, device="cuda")
        plan.run(Yd, nrow, out)
        torch.cuda.synchronize()
        assert_equal(out.cpu().numpy().T, want, tol=1e-9, atol=1e-11, what=f"{case} poison={poison}")


def test_pbc_dma_path_ragged_columns(hip, oracle):
    """Empty columns, one very long column, empty row ranges (empty tiles)."""
    from sparsearray_amd.device import PbcPlan
    nrow, ncol, K = 40000, 530, 128
    rng = np.random.default_rng(33)
    dense = np.zeros((nrow, ncol))
    dense[:, 7] = rng.uniform(-1, 1, nrow)                       # full column
    for c in range(0, ncol, 3):
        rows = rng.integers(20000, 26000, 40)                    # a band: most tiles empty
        dense[rows, c] = rng.uniform(-1, 1, 40)
    dense[nrow - 1, ncol - 1] = 2.5
    x = SVT_SparseArray.from_dense(dense)
    cp, ri, v = x.to_csc()
    y = rng.uniform(-1, 1, (nrow, K))
    want = oracle.crossprod(x, y)
    A = _dev(cp, ri, v, nrow)
    Yd = torch.as_tensor(np.ascontiguousarray(y.T), device="cuda")
    out = torch.zeros((K, ncol), dtype=torch.float64, device="cuda")
    PbcPlan(A, K, 40, 16, 7).run(Yd, nrow, out)
    torch.cuda.synchronize()
    assert_equal(out.cpu().numpy().T, want, tol=1e-9, atol=1e-11, what="ragged")


@pytest.mark.parametrize("shape", [(3000, 700, 0.01), (257, 5, 0.3), (40, 20000, 0.002), (1, 9, 1.0),
                                   (5003, 3000, 0.05), (17, 4000, 0.5), (2_100_000, 12, 0.0005)])
@pytest.mark.parametrize("dtype", ["double", "integer"])
def test_device_transpose(hip, shape, dtype):
    """t(A) on the device == t() of the host mirror (leaf entries in ascending order)."""
    nrow, ncol, dens = shape
    cp, ri, v = random_csc(nrow, ncol, dens, seed=41)
    if dtype == "integer":
        v = np.round(v * 100).astype(np.int32)
    x = SVT_SparseArray.from_csc((nrow, ncol), dtype, cp, ri, v)
    tcp, tri, tv = x.t().to_csc()
    A = _dev(cp, ri, v, nrow)
    T = A.t()
    torch.cuda.synchronize()
    assert (T.nrow, T.ncol, T.nnz) == (ncol, nrow, len(v))
    assert np.array_equal(T.col_ptr.cpu().numpy(), tcp)
    assert np.array_equal(T.row_idx.cpu().numpy(), tri)
    assert np.array_equal(T.val.cpu().numpy(), tv)


def test_device_transpose_skewed_rows(hip):
    """Rows of very different lengths in one bucket of 16 (the two-pass form of the transposition finishes
    the order inside such a bucket with one wavefront): a full row, an empty row and sparse rows side by
    side, plus whole empty buckets."""
    rng = np.random.default_rng(5)
    nrow, ncol = 1000, 2500
    m = np.zeros((nrow, ncol))
    m[35, :] = rng.normal(size=ncol)                      # full row
    m[37, ::3] = 1.5
    m[40:48, :] = np.where(rng.random((8, ncol)) < 0.02, 2.0, 0.0)
    m[999, 7] = -1.0                                      # rows 48..998 empty
    x = SVT_SparseArray.from_dense(m, type="double", lacunar=False)
    cp, ri, v = x.to_csc()
    tcp, tri, tv = x.t().to_csc()
    T = _dev(cp, ri, v, nrow).t()
    torch.cuda.synchronize()
    assert np.array_equal(T.col_ptr.cpu().numpy(), tcp)
    assert np.array_equal(T.row_idx.cpu().numpy(), tri)
    assert np.array_equal(T.val.cpu().numpy(), tv)


@pytest.mark.parametrize("perm", [(3, 2, 1), (1, 3, 2), (2, 3, 1), (3, 1, 2), (2, 1, 3)])
def test_device_aperm(hip, perm):
    """aperm() of a 3-d array held in the device layout == numpy's transpose of the dense array."""
    dim = (700, 40, 23)
    rng = np.random.default_rng(44)
    a = np.zeros(dim, order="F")
    idx = rng.choice(a.size, size=20000, replace=False)
    a.reshape(-1, order="F")[idx] = rng.normal(size=idx.size)
    x = SVT_SparseArray.from_dense(a, "double", lacunar=False)
    cp, ri, v = x.to_csc()
    A = _dev(cp, ri, v, dim[0])
    T, new_dim = A.aperm(dim, perm)
    torch.cuda.synchronize()
    want = SVT_SparseArray.from_dense(np.asfortranarray(np.transpose(a, [p - 1 for p in perm])), "double",
                                      lacunar=False)
    wcp, wri, wv = want.to_csc()
    assert new_dim == want.dim
    assert np.array_equal(T.col_ptr.cpu().numpy(), wcp)
    assert np.array_equal(T.row_idx.cpu().numpy(), wri)
    assert np.array_equal(T.val.cpu().numpy(), wv)


@pytest.mark.parametrize("dim,nnz,perm,dtype", [
    ((3000, 2500, 5), 750_000, (2, 1, 3), "double"),       # F = 16 rows, 10 groups of 256 columns (the last of 196)
    ((3000, 2500, 5), 750_000, (2, 1, 3), "integer"),
    ((1234, 777, 3, 2), 400_000, (2, 1, 3, 4), "double"),  # 4-d: six slabs, ragged groups and buckets
    ((5000, 300, 7), 900_000, (2, 1, 3), "double"),        # tall slabs
    ((700, 40, 23), 20000, (2, 1, 3), "double"),           # too few nonzeros per column and bucket: the key sort
    # 3-d: c(2,3,1) = c(2,1,3) then c(1,3,2), c(3,2,1) = c(2,1,3) then c(3,1,2) through an intermediate array
    ((3000, 2500, 5), 750_000, (2, 3, 1), "double"),
    ((3000, 2500, 5), 750_000, (3, 2, 1), "double"),
    ((3000, 2500, 5), 750_000, (3, 2, 1), "integer"),
    ((5000, 300, 7), 900_000, (2, 3, 1), "integer"),
])
def test_device_aperm_first_two_axes_swapped(hip, dim, nnz, perm, dtype):
    """aperm(x, c(2, 1, 3, ...)): every slab of the remaining axes is a matrix transposed on its own -- the bucketed
    transposition, batched over the slabs (no library sort; src/SparseArray_aperm.c:892-929 in the reference) --
    against numpy; one slab is emptied, one column of another holds no nonzero."""
    rng = np.random.default_rng(46)
    a = np.zeros(dim, order="F")
    idx = rng.choice(a.size, size=nnz, replace=False)
    a.reshape(-1, order="F")[idx] = rng.normal(size=nnz) if dtype == "double" else rng.integers(1, 1000, size=nnz)
    a[(slice(None), slice(None)) + (1,) + (0,) * (len(dim) - 3)] = 0      # an empty slab
    a[(slice(None), 7) + (0,) * (len(dim) - 2)] = 0                       # an empty column
    x = SVT_SparseArray.from_dense(a, dtype, lacunar=False)
    cp, ri, v = x.to_csc()
    A = _dev(cp, ri, v, dim[0])
    T, new_dim = A.aperm(dim, perm)
    torch.cuda.synchronize()
    want = SVT_SparseArray.from_dense(np.asfortranarray(np.transpose(a, [q - 1 for q in perm])), dtype, lacunar=False)
    wcp, wri, wv = want.to_csc()
    assert new_dim == want.dim
    assert np.array_equal(T.col_ptr.cpu().numpy(), wcp)
    assert np.array_equal(T.row_idx.cpu().numpy(), wri)
    assert np.array_equal(T.val.cpu().numpy(), wv)


@pytest.mark.parametrize("dim,nnz,perm,dtype", [
    ((700, 40, 23), 20000, (3, 1, 2), "integer"),          # slabs of ~500 nonzeros
    ((5000, 9, 64), 30000, (3, 1, 2), "double"),           # ~3300 per slab, 64 old leaves each
    ((300, 6, 5, 4), 9000, (3, 1, 2, 4), "double"),        # 4-d: slabs over two remaining axes
    ((300, 6, 5, 4), 9000, (4, 1, 3, 2), "double"),        #      ... taken in another order
    ((64, 50, 1), 1500, (3, 1, 2), "double"),              # one entry along the new leading axis
    ((20000, 3, 64), 800000, (3, 1, 2), "double"),         # slabs over the cap: the key sort takes over
    ((900, 30, 16), 0, (3, 1, 2), "double"),               # no nonzeros
])
def test_device_aperm_slab_form(hip, dim, nnz, perm, dtype):
    """Permutations whose new leading axis is an old outer axis and whose second axis is the old rows run one
    workgroup per slab (aperm_slab_kernel, src/SparseArray_aperm.c:892-929 in the reference): against numpy."""
    rng = np.random.default_rng(45)
    a = np.zeros(dim, order="F")
    if nnz:
        idx = rng.choice(a.size, size=nnz, replace=False)
        vals = rng.normal(size=nnz) if dtype == "double" else rng.integers(1, 1000, size=nnz)
        a.reshape(-1, order="F")[idx] = vals
    if dtype == "integer":
        a = a.astype(np.int32)
    x = SVT_SparseArray.from_dense(a, dtype, lacunar=False)
    cp, ri, v = x.to_csc()
    A = _dev(cp, ri, v, dim[0])
    T, new_dim = A.aperm(dim, perm)
    torch.cuda.synchronize()
    want = SVT_SparseArray.from_dense(np.asfortranarray(np.transpose(a, [p - 1 for p in perm])), dtype,
                                      lacunar=False)
    wcp, wri, wv = want.to_csc()
    assert new_dim == want.dim
    assert np.array_equal(T.col_ptr.cpu().numpy(), wcp)
    assert np.array_equal(T.row_idx.cpu().numpy(), wri)
    assert np.array_equal(T.val.cpu().numpy(), wv)


@pytest.mark.parametrize("dim,nnz,perm,dtype", [
    # general permutations of arrays with four and five axes (round 5): a leaf-preserving step, the first two axes
    # swapped (batched bucketed transposition), a leaf-preserving step -- no sort
    ((1500, 900, 4, 3), 1_500_000, (2, 4, 1, 3), "double"),     # q = 2: no first step
    ((1500, 900, 4, 3), 1_500_000, (3, 1, 4, 2), "double"),     # dim[2] = 4: slab form refused? (perm[1] == 1: slab form takes it)
    ((1500, 4, 900, 3), 1_500_000, (3, 2, 4, 1), "double"),     # q = 3: all three steps
    ((1500, 4, 900, 3), 1_500_000, (3, 4, 2, 1), "integer"),
    ((1200, 5, 3, 700, 2), 1_200_000, (4, 5, 1, 3, 2), "double"),   # five axes
    ((1200, 5, 3, 700, 2), 1_200_000, (4, 1, 2, 3, 5), "double"),   # q = 4, the rest in order after the swap? (no last step)
    # shapes the bucketed transposition refuses: the library's own radix sort of (new leaf, position) pairs
    ((300, 6, 5, 4), 9000, (2, 4, 3, 1), "double"),
    ((300, 6, 5, 4), 9000, (4, 3, 2, 1), "integer"),
    ((40, 30, 20, 10, 3), 50_000, (5, 3, 1, 4, 2), "double"),
])
def test_device_aperm_general_permutations(hip, dim, nnz, perm, dtype):
    """aperm() of 4-d / 5-d arrays for permutations that are none of the special forms, against numpy's transpose
    (C_aperm_SVT, src/SparseArray_aperm.c:892-970)."""
    rng = np.random.default_rng(47)
    a = np.zeros(dim, order="F")
    idx = rng.choice(a.size, size=nnz, replace=False)
    a.reshape(-1, order="F")[idx] = rng.normal(size=nnz) if dtype == "double" else rng.integers(1, 1000, size=nnz)
    if dtype == "integer":
        a = a.astype(np.int32)
    x = SVT_SparseArray.from_dense(a, dtype, lacunar=False)
    cp, ri, v = x.to_csc()
    A = _dev(cp, ri, v, dim[0])
    T, new_dim = A.aperm(dim, perm)
    torch.cuda.synchronize()
    want = SVT_SparseArray.from_dense(np.asfortranarray(np.transpose(a, [q - 1 for q in perm])), dtype, lacunar=False)
    wcp, wri, wv = want.to_csc()
    assert new_dim == want.dim
    assert np.array_equal(T.col_ptr.cpu().numpy(), wcp)
    assert np.array_equal(T.row_idx.cpu().numpy(), wri)
    assert np.array_equal(T.val.cpu().numpy(), wv)


def test_transpose_shapes_for_the_own_radix_sort(hip):
    """t() of operands the bucketed transposition does not take (less than one nonzero per column and coarse bucket):
    the library's own least-significant-digit radix sort (svt_sort.h; rounds 1-4: rocprim) -- one, two, three and four
    passes of 8 bits over the row index, a ragged last tile, empty columns; against a stable host sort."""
    from sparsearray_amd.device import DeviceCSC
    rng = np.random.default_rng(48)
    for nrow, ncol, nnz in ((200, 70_000, 30_000), (60_000, 50_000, 41_000), (3_000_000, 9_000, 10_000),
                            (20_000_000, 3_000, 70_001), (100, 5, 3)):
        lin = np.sort(rng.choice(nrow * ncol, size=nnz, replace=False))
        col, row = lin // nrow, (lin % nrow).astype(np.int32)
        cp = np.zeros(ncol + 1, dtype=np.int64)
        np.add.at(cp, col + 1, 1)
        cp = np.cumsum(cp)
        v = rng.normal(size=nnz)
        T = DeviceCSC.from_host(nrow, cp, row, v).t()
        torch.cuda.synchronize()
        order = np.argsort(row, kind="stable")
        tcp = np.zeros(nrow + 1, dtype=np.int64)
        np.cumsum(np.bincount(row, minlength=nrow), out=tcp[1:])
        assert np.array_equal(T.col_ptr.cpu().numpy(), tcp), (nrow, ncol)
        assert np.array_equal(T.row_idx.cpu().numpy(), col[order].astype(np.int32)), (nrow, ncol)
        assert np.array_equal(T.val.cpu().numpy(), v[order]), (nrow, ncol)


def test_device_matmul_through_transpose(hip, oracle):
    """x %*% y = crossprod(t(x), y) (R/SparseMatrix-mult.R:195-215) with everything on the
    device: transpose, panel-blocked layout of t(x), product; many column blocks, one row split."""
    from sparsearray_amd.device import PbcPlan
    nrow, ncol, K = 60000, 900, 64
    cp, ri, v = random_csc(nrow, ncol, 0.01, seed=42)
    v = v.copy()
    v[7] = NA_real
    x = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, v)
    y = np.random.default_rng(43).uniform(-1, 1, (ncol, K))
    want = oracle.matmul(x, y)                                          # nrow x K
    T = _dev(cp, ri, v, nrow).t()                                       # ncol x nrow
    Yd = torch.as_tensor(np.ascontiguousarray(y.T), device="cuda")      # (K, ncol)
    out = torch.zeros((K, nrow), dtype=torch.float64, device="cuda")
    PbcPlan(T, K).run(Yd, ncol, out)
    torch.cuda.synchronize()
    assert_equal(out.cpu().numpy().T, want, tol=1e-9, atol=1e-11, strict_na=True, what="matmul")


def test_device_stats(hip, oracle):
    from sparsearray_amd.device import colstats, rowsum, rowsums
    nrow, ncol = 30000, 64
    cp, ri, v = random_csc(nrow, ncol, 0.05, seed=25)
    x = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, v)
    A = _dev(cp, ri, v, nrow)
    for op, fn in (("sum", oracle.colSums), ("mean", oracle.colMeans), ("var1", oracle.colVars)):
        got, _ = colstats(A, op)
        assert_equal(got.cpu().numpy(), fn(x), tol=1e-6, atol=1e-9, what=op)
    assert_equal(rowsums(A).cpu().numpy(), oracle.rowSums(x), tol=1e-6, atol=1e-9)
    grp = np.random.default_rng(26).integers(1, 11, nrow).astype(np.int32)
    got = rowsum(A, torch.as_tensor(grp, device="cuda"), 10).cpu().numpy().T
    want, _ = oracle.SparseArray_Call("C_rowsum_SVT", x, grp, 10, False)
    assert_equal(got, want, tol=1e-6, atol=1e-9)


@pytest.mark.parametrize("na_rm", [False, True])
def test_device_rowsum_long_columns_16bit_group_table(hip, na_rm):
    """nrow >= 65536 and < 65535 groups: the kernel gathers from a 16-bit copy of the group table
    (kernels_rowstats.hip); NA groups land in the last slot (src/rowsum_methods.c:44-64)."""
    from sparsearray_amd import NA_integer
    from sparsearray_amd.device import rowsum
    nrow, ncol, ngroup = 100_000, 60, 700
    cp, ri, v = random_csc(nrow, ncol, 0.05, seed=41)
    v = v.copy()
    v[::97] = np.nan
    A = _dev(cp, ri, v, nrow)
    rng = np.random.default_rng(42)
    grp = rng.integers(1, ngroup, nrow).astype(np.int32)       # 1 .. ngroup-1
    grp[rng.integers(0, nrow, 500)] = NA_integer                # NA group = slot ngroup
    out = rowsum(A, torch.as_tensor(grp, device="cuda"), ngroup, na_rm=na_rm)
    torch.cuda.synchronize()
    got = out.cpu().numpy()                                      # (ncol, ngroup)
    g0 = np.where(grp == NA_integer, ngroup, grp) - 1
    want = np.zeros((ncol, ngroup))
    for j in range(ncol):
        rows, vals = ri[cp[j]:cp[j + 1]], v[cp[j]:cp[j + 1]]
        if na_rm:
            keep = ~np.isnan(vals)
            rows, vals = rows[keep], vals[keep]
        np.add.at(want[j], g0[rows], vals)
    assert_equal(got, want, tol=1e-12, atol=1e-12, what="rowsum g16")


def test_device_colmedians(hip):
    from sparsearray_amd.device import colmedians
    nrow, ncol = 30_000, 200
    cp, ri, v = random_csc(nrow, ncol, 0.6, seed=51)          # dense enough for non-zero medians
    A = _dev(cp, ri, v, nrow)
    got = colmedians(A).cpu().numpy()
    want = np.empty(ncol)
    for j in range(ncol):
        col = np.zeros(nrow)
        col[ri[cp[j]:cp[j + 1]]] = v[cp[j]:cp[j + 1]]
        want[j] = np.median(col)
    assert_equal(got, want, tol=1e-15, what="colmedians")


@pytest.mark.parametrize("nrow", [2000, 38480, 65 * 128 + 3])
def test_pbc_dma_partial_last_panel_first_in_a_split(hip, oracle, nrow):
    """Row splits whose in-loop staging starts with the partial last panel: its window is moved back
    by less than one panel, and the kernel's staging offset register is unsigned -- a fault at 300000
    columns x 2000 rows once (svt_matmul_SVT_SVT).  Shapes for which the launcher's own choice of
    row splits puts the partial panel first in its split: 2000 rows = 16 splits of one panel (the last
    of 80 rows); 38480 rows = 301 panels in splits of 3 (the last split is the 80-row panel alone);
    8323 rows = 66 panels, the last of 3 rows."""
    from sparsearray_amd.device import PbcPlan
    ncol, K = 900, 50
    cp, ri, v = random_csc(nrow, ncol, 0.02 if nrow < 10000 else 0.004, seed=61)
    x = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, v)
    y = np.random.default_rng(62).uniform(-1, 1, (nrow, K))
    want = oracle.crossprod(x, y)
    A = _dev(cp, ri, v, nrow)
    Yd = torch.as_tensor(np.ascontiguousarray(y.T), device="cuda")
    plan = PbcPlan(A, K)
    out = torch.zeros((K, ncol), dtype=torch.float64, device="cuda")
    plan.run(Yd, nrow, out)
    torch.cuda.synchronize()
    assert_equal(out.cpu().numpy().T, want, tol=1e-9, atol=1e-11, what=f"nrow {nrow}")


def test_pbc_dirty_columns_fixed_up_per_column(hip, oracle):
    """Non-finite entries of the dense operand: only their columns are rewritten (the reference's
    per-column switch to the slow dot product, src/SparseMatrix_mult.c:193-207).  Entries on rows
    where some leaves have a nonzero (those leaves keep the IEEE sum: Inf, or NaN from Inf - Inf)
    and on rows where they have none (NaN from 0 * Inf), an R NA (the whole column NA), leaves that
    hold an NA themselves, several entries in one column, and the clean columns bit for bit."""
    from sparsearray_amd.device import PbcPlan
    nrow, ncol, K = 30000 + 50, 1500, 70
    cp, ri, v = random_csc(nrow, ncol, 0.01, seed=71)
    v = v.copy()
    v[cp[17] + 3] = NA_real                              # leaf 17 holds an R NA
    x = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, v)
    A = _dev(cp, ri, v, nrow)
    plan = PbcPlan(A, K)
    rng = np.random.default_rng(72)
    y0 = rng.uniform(-1, 1, (nrow, K))
    out = torch.zeros((K, ncol), dtype=torch.float64, device="cuda")

    def run(y):
        plan.run(torch.as_tensor(np.ascontiguousarray(y.T), device="cuda"), nrow, out)
        torch.cuda.synchronize()
        return out.cpu().numpy().T.copy()

    clean = run(y0)
    y = y0.copy()
    r_hit = int(ri[cp[5] + 2])                          # a row where leaf 5 has a nonzero
    r_hit2 = int(ri[cp[5] + 7])
    y[r_hit, 3] = np.inf                                # column 3: one entry, hit by leaf 5 (and others)
    y[r_hit, 9] = np.inf; y[r_hit2, 9] = -np.inf        # column 9: two entries, both hit by leaf 5
    y[12345, 20] = np.nan                               # column 20: a NaN
    y[nrow - 1, 69] = NA_real                           # column 69: an R NA in the last partial panel
    y[0, 0] = -np.inf
    got = run(y)
    want = oracle.crossprod(x, y)
    assert_equal(got, want, tol=1e-9, atol=1e-11, strict_na=True, what="dirty columns")
    assert np.isinf(got[5, 3]) and (np.isnan(got[6, 3]) or np.isinf(got[6, 3]))
    dirty = [0, 3, 9, 20, 69]
    keep = [k for k in range(K) if k not in dirty]
    assert np.array_equal(got[:, keep], clean[:, keep], equal_nan=True)     # untouched, bit for bit
    # a clean operand afterwards: nothing sticks
    assert np.array_equal(run(y0), clean, equal_nan=True)
    # a whole column of NaN (more entries than the fix-up lists): the general kernels take over
    y = y0.copy()
    y[:, 11] = np.nan
    y[77, 12] = np.inf
    assert_equal(run(y), oracle.crossprod(x, y), tol=1e-9, atol=1e-11, strict_na=True, what="NaN column")
    # more dirty columns than the fix-up handles
    y = y0.copy()
    y[100, ::3] = np.inf
    assert_equal(run(y), oracle.crossprod(x, y), tol=1e-9, atol=1e-11, strict_na=True, what="many dirty columns")


@pytest.mark.parametrize("shape", [(20000 + 77, 2100, 70), (4096, 700, 64), (1282, 90, 130)])
def test_pbc_dense_operand_given_by_rows(hip, oracle, shape):
    """tr_y (tcrossprod / transpose.x, src/SparseMatrix_mult.c:411-421): the K x nrow operand is
    transposed on the device and takes the same product kernel: bit-identical to the column-major
    call, and equal to the oracle; with a non-finite entry too."""
    from sparsearray_amd.device import PbcPlan
    nrow, ncol, K = shape
    cp, ri, v = random_csc(nrow, ncol, 0.01, seed=81)
    x = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, v)
    A = _dev(cp, ri, v, nrow)
    plan = PbcPlan(A, K)
    y = np.random.default_rng(82).uniform(-1, 1, (nrow, K))
    for poison in (False, True):
        if poison:
            y[nrow // 2, K - 1] = np.inf
        Ycm = torch.as_tensor(np.ascontiguousarray(y.T), device="cuda")     # (K, nrow) = column-major nrow x K
        Yrm = torch.as_tensor(np.ascontiguousarray(y), device="cuda")       # (nrow, K) = column-major K x nrow
        o1 = torch.zeros((K, ncol), dtype=torch.float64, device="cuda")
        o2 = torch.zeros((K, ncol), dtype=torch.float64, device="cuda")
        plan.run(Ycm, nrow, o1)
        plan.run(Yrm, K, o2, tr_y=True)
        torch.cuda.synchronize()
        assert torch.equal(torch.nan_to_num(o1, nan=7.0), torch.nan_to_num(o2, nan=7.0))
        assert_equal(o2.cpu().numpy().T, oracle.crossprod(x, y), tol=1e-9, atol=1e-11, strict_na=True,
                     what=f"tr_y poison={poison}")


@pytest.mark.parametrize("shape", [(200_000, 3000, 128, 0.001), (70_000 + 31, 1100, 70, 0.002), (5000, 170, 64, 0.004)])
def test_pbc_gather_kernel_matches_oracle(hip, oracle, shape):
    """Very sparse operands: layout (40, 4, 10) and crossprod_pbc_gather_kernel (rows of the
    row-major copy of Y straight from L2, no LDS staging).  Both orientations of the dense operand,
    an NA in a leaf, non-finite entries in Y (per-column fix-up on the row-major copy)."""
    from sparsearray_amd.device import PbcPlan
    nrow, ncol, K, dens = shape
    cp, ri, v = random_csc(nrow, ncol, dens, seed=91)
    v = v.copy()
    v[cp[ncol // 2] if cp[ncol // 2] < len(v) else 0] = NA_real
    x = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, v)
    A = _dev(cp, ri, v, nrow)
    plan = PbcPlan(A, K, 40, 4, 10)
    y = np.random.default_rng(92).uniform(-1, 1, (nrow, K))
    out = torch.zeros((K, ncol), dtype=torch.float64, device="cuda")
    for poison in (0, 1, 2):
        if poison == 1:
            y[int(ri[cp[3]]) if cp[4] > cp[3] else 5, K - 1] = np.inf     # on a nonzero of leaf 3
            y[nrow - 1, 0] = np.nan
        if poison == 2:
            y[:, 2] = np.nan                                               # general kernels take over
        want = oracle.crossprod(x, y)
        plan.run(torch.as_tensor(np.ascontiguousarray(y.T), device="cuda"), nrow, out)
        torch.cuda.synchronize()
        assert_equal(out.cpu().numpy().T, want, tol=1e-9, atol=1e-11, strict_na=True, what=f"poison {poison}")
        out.fill_(5.0)
        plan.run(torch.as_tensor(np.ascontiguousarray(y), device="cuda"), K, out, tr_y=True)
        torch.cuda.synchronize()
        assert_equal(out.cpu().numpy().T, want, tol=1e-9, atol=1e-11, strict_na=True, what=f"by rows, poison {poison}")


@pytest.mark.parametrize("cfg", [(40, 11), (32, 10), (16, 9)], ids=["40x2048", "32x1024", "16x512"])
def test_pbc_gather_xcd_paced_kernel_matches_oracle(hip, oracle, cfg):
    """crossprod_pbc_gatherx_kernel (persistent grid, one row range per XCD, wavefronts paced through
    progress words; K = 128, >= 64 panels): against the oracle for the three accumulator widths, a ragged
    last panel, an empty leaf block at the end of an XCD's range, CUs left idle, and bit for bit against
    itself with the pacing switched off (no result depends on the protocol) -- the reference's loop is
    src/SparseMatrix_mult.c:131-152 over src/SparseVec_dotprod.c:28-43."""
    from sparsearray_amd.device import PbcPlan, set_gather_pacing, set_spare_cus
    cbw, logr = cfg
    nrow, ncol, K = (64 << logr) + 777, 1500 + cbw + 3, 128
    cp, ri, v = random_csc(nrow, ncol, 0.0015, seed=191)
    keep = ri < nrow - (3 << logr)                      # the last XCD's last panels hold no nonzero at all
    kept = np.concatenate([[0], np.cumsum(keep)]).astype(np.int64)
    ri, v, cp = ri[keep], v[keep], kept[cp]
    x = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, v)
    A = _dev(cp, ri, v, nrow)
    plan = PbcPlan(A, K, cbw, 4, logr)
    y = np.random.default_rng(192).uniform(-1, 1, (nrow, K))
    Yd = torch.as_tensor(np.ascontiguousarray(y.T), device="cuda")
    want = oracle.crossprod(x, y)
    outs = []
    try:
        for dsync, spare in ((2, 0), (0, 0), (1_000_000, 0), (2, 96), (-1, 0)):
            set_gather_pacing(dsync, 64)
            set_spare_cus(spare)
            out = torch.full((K, ncol), 3.0, dtype=torch.float64, device="cuda")
            plan.run(Yd, nrow, out)
            torch.cuda.synchronize()
            assert_equal(out.cpu().numpy().T, want, tol=1e-9, atol=1e-11, what=f"dsync {dsync} spare {spare}")
            outs.append(out)
    finally:
        set_gather_pacing()
        set_spare_cus(0)
    # same kernel, same order of additions per cell, whatever the pacing
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_pbc_gather_xcd_paced_wide_dense_operand_nonfinite_high_column(hip, oracle):
    """K = 1024 through the paced gather kernel with an Inf (no NA) in dense columns past 992: the per-column
    counters of the non-finite fix-up must not share bytes with the kernel's progress words (ADVICE round 4:
    with the counters inside the flag block, has_na[k] for k >= 993 sat on the progress words and the fix-up
    wrote NA_real_ where the reference gives NaN / Inf; src/SparseVec_dotprod.c:48-65)."""
    from sparsearray_amd.device import PbcPlan
    logr, K = 9, 1024
    nrow, ncol = (64 << logr) + 100, 333
    cp, ri, v = random_csc(nrow, ncol, 0.002, seed=291)
    x = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, v)
    A = _dev(cp, ri, v, nrow)
    plan = PbcPlan(A, K, 40, 4, logr)
    y = np.random.default_rng(292).uniform(-1, 1, (nrow, K))
    y[int(ri[cp[7]]), 1000] = np.inf           # on a nonzero of leaf 7
    y[11, 1023] = -np.inf
    y[nrow - 1, 995] = np.nan
    y[5, 3] = np.inf
    out = torch.full((K, ncol), 3.0, dtype=torch.float64, device="cuda")
    plan.run(torch.as_tensor(np.ascontiguousarray(y.T), device="cuda"), nrow, out)
    torch.cuda.synchronize()
    assert_equal(out.cpu().numpy().T, oracle.crossprod(x, y), tol=1e-9, atol=1e-11, strict_na=True, what="K = 1024")


def test_pbc_many_column_blocks_last_round_cut_by_rows(hip, oracle):
    """A product with many column blocks and no row split (the shape of A %*% Y on the layout of t(A), BASELINE config 2b):
    one launch per round of workgroups, and the last, partly filled round cut by rows with its partial sums added in
    split order (round 5).  Against the oracle, against the single launch, with a short last column block, a ragged
    last panel, an NA in a leaf of the last round and non-finite entries in Y (the fix-up reads the flags of all launches);
    src/SparseMatrix_mult.c:131-152 over src/SparseVec_dotprod.c:28-65."""
    from sparsearray_amd.device import PbcPlan, set_round_launches
    nrow, K = 8192 + 70, 128
    ncol = (128 * 2 + 27) * 640 - 100
    cp, ri, v = random_csc(nrow, ncol, 0.002, seed=391)
    v = v.copy()
    c_na, c_inf = ncol - 8000, ncol - 7000                    # leaves of the last round (random_csc() leaves the last
    assert cp[c_na + 1] > cp[c_na] and cp[c_inf + 1] > cp[c_inf]   # ~2 % of the columns of a large operand empty)
    v[cp[c_na]] = NA_real
    x = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, v)
    A = _dev(cp, ri, v, nrow)
    plan = PbcPlan(A, K)
    y = np.random.default_rng(392).uniform(-1, 1, (nrow, K))
    outs = {}
    try:
        for poison in (0, 1):
            if poison:
                y[int(ri[cp[c_inf]]), 7] = np.inf             # on a nonzero of a leaf of the last round
                y[nrow - 1, 100] = np.nan
            Yd = torch.as_tensor(np.ascontiguousarray(y.T), device="cuda")
            want = oracle.crossprod(x, y)
            for mode in (1, 2, 0):
                set_round_launches(mode)
                out = torch.full((K, ncol), 3.0, dtype=torch.float64, device="cuda")
                plan.run(Yd, nrow, out)
                torch.cuda.synchronize()
                assert_equal(out.cpu().numpy().T, want, tol=1e-9, atol=1e-11, strict_na=True, what=f"mode {mode} poison {poison}")
                outs[(poison, mode)] = out
            # rounds do not change a sum; the row-split round adds the same products in another order
            assert torch.equal(outs[(poison, 2)].nan_to_num(1.5), outs[(poison, 0)].nan_to_num(1.5))
    finally:
        set_round_launches(1)


def test_pbc_flags_between_products(hip, oracle):
    """The flags and per-column counters of the non-finite fix-up live at the head of the workspace and are cleared by one
    small kernel in front of every product.  Sequences that would show a stale flag: a non-finite product followed by a
    clean one, a phase 1 whose phase 2 never came, two streams taking turns on one handle with a dirty and a clean
    operand.  (Round 5 also built flag blocks of the library's own, kept zero between products so that the clear kernel
    could go: all tests green, but the step at an eighth of the rows of config 2a did not get shorter -- 267.8 -> 267.0 us
    on the kernel timeline: the product kernel's start absorbs what the 5 us kernel in front of it took -- dropped.)"""
    from sparsearray_amd.device import PbcPlan
    nrow, ncol, K = 30000, 1300, 128
    cp, ri, v = random_csc(nrow, ncol, 0.01, seed=501)
    x = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, v)
    A = _dev(cp, ri, v, nrow)
    plan = PbcPlan(A, K)
    y = np.random.default_rng(502).uniform(-1, 1, (nrow, K))
    yd = y.copy(); yd[int(ri[cp[11]]), 3] = np.inf; yd[17, 77] = np.nan; yd[:, 100] = NA_real
    Yc = torch.as_tensor(np.ascontiguousarray(y.T), device="cuda")
    Yd = torch.as_tensor(np.ascontiguousarray(yd.T), device="cuda")
    want_c, want_d = oracle.crossprod(x, y), oracle.crossprod(x, yd)
    out = torch.zeros((K, ncol), dtype=torch.float64, device="cuda")

    def check(Y, want, what):
        out.fill_(7.0)
        plan.run(Y, nrow, out)
        torch.cuda.synchronize()
        assert_equal(out.cpu().numpy().T, want, tol=1e-9, atol=1e-11, strict_na=True, what=what)
    for rep in range(3):
        check(Yd, want_d, f"dirty {rep}")
        check(Yc, want_c, f"clean after dirty {rep}")
    plan.run_phase(1, Yd, nrow, out)                    # a phase 1 without its phase 2 ...
    check(Yc, want_c, "clean after an orphaned dirty phase 1")
    check(Yd, want_d, "dirty again")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    o1 = torch.zeros_like(out); o2 = torch.zeros_like(out)
    for rep in range(4):
        with torch.cuda.stream(s1):
            plan.run(Yd if rep % 2 == 0 else Yc, nrow, o1)
        torch.cuda.synchronize()                        # (one plan = one workspace of partial sums: not both at once)
        with torch.cuda.stream(s2):
            plan.run(Yc if rep % 2 == 0 else Yd, nrow, o2)
        torch.cuda.synchronize()
        a, b = (want_d, want_c) if rep % 2 == 0 else (want_c, want_d)
        assert_equal(o1.cpu().numpy().T, a, tol=1e-9, atol=1e-11, strict_na=True, what=f"stream 1, turn {rep}")
        assert_equal(o2.cpu().numpy().T, b, tol=1e-9, atol=1e-11, strict_na=True, what=f"stream 2, turn {rep}")


def test_pbc_auto_layout_picks_by_density(hip, oracle):
    """svt_dev_pbc_build(A, 0, 0, 0): the gather layout below ~0.25 % density, the LDS-DMA layout
    above; same results either way."""
    from sparsearray_amd.device import PbcPlan
    for dens in (0.0008, 0.01):
        nrow, ncol, K = 60_000, 900, 64
        cp, ri, v = random_csc(nrow, ncol, dens, seed=95)
        x = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, v)
        A = _dev(cp, ri, v, nrow)
        y = np.random.default_rng(96).uniform(-1, 1, (nrow, K))
        Yd = torch.as_tensor(np.ascontiguousarray(y.T), device="cuda")
        o_auto = torch.zeros((K, ncol), dtype=torch.float64, device="cuda")
        PbcPlan(A, K, 0, 0, 0).run(Yd, nrow, o_auto)
        torch.cuda.synchronize()
        assert_equal(o_auto.cpu().numpy().T, oracle.crossprod(x, y), tol=1e-9, atol=1e-11, what=f"density {dens}")


def test_pbc_dirty_column_classes(hip, oracle):
    """A whole column of NaN / NA in the dense operand (more non-finite entries than the longest leaf has
    nonzeros: every cell NaN or NA, decided without listing the entries), a column with more entries than
    the fix-up lists but fewer than the longest leaf (general kernels), and both next to a single Inf."""
    from sparsearray_amd.device import PbcPlan
    nrow, ncol, K = 20000, 700, 64
    cp, ri, v = random_csc(nrow, ncol, 0.01, seed=101)
    # leaf 9 becomes long: 4000 nonzeros (rows 0 .. 3999)
    rows9 = np.arange(4000, dtype=np.int32)
    vals9 = np.random.default_rng(102).uniform(-1, 1, 4000)
    ri = np.concatenate([ri[:cp[9]], rows9, ri[cp[10]:]])
    v = np.concatenate([v[:cp[9]], vals9, v[cp[10]:]])
    shift = 4000 - (cp[10] - cp[9])
    cp = cp.copy(); cp[10:] += shift
    x = SVT_SparseArray.from_csc((nrow, ncol), "double", cp, ri, v)
    A = _dev(cp, ri, v, nrow)
    plan = PbcPlan(A, K)
    y0 = np.random.default_rng(103).uniform(-1, 1, (nrow, K))
    out = torch.zeros((K, ncol), dtype=torch.float64, device="cuda")

    def check(y, what):
        plan.run(torch.as_tensor(np.ascontiguousarray(y.T), device="cuda"), nrow, out)
        torch.cuda.synchronize()
        assert_equal(out.cpu().numpy().T, oracle.crossprod(x, y), tol=1e-9, atol=1e-11, strict_na=True, what=what)

    y = y0.copy(); y[:, 5] = np.nan; y[77, 6] = np.inf
    check(y, "NaN column (saturated) + one Inf")
    y = y0.copy(); y[:, 5] = NA_real; y[:, 40] = np.inf; y[3, 41] = -np.inf
    check(y, "NA column + Inf column (both saturated) + one -Inf")
    y = y0.copy(); y[:3000, 7] = np.inf                       # 3000 entries: listed 256, longest leaf 4000 -> general kernels
    check(y, "between the list and the longest leaf")
    y = y0.copy(); y[:3000, 7] = np.inf; y[:, 8] = np.nan
    check(y, "in-between column beside a saturated one")
    # round 5: every class is decided inside the leaf kernel (no general kernels behind the panel products)
    rng = np.random.default_rng(104)
    y = y0.copy()
    for k in range(40):                                       # 40 light columns (round 4: more than 16 -> general kernels)
        y[rng.integers(0, nrow, 3), k] = [np.inf, -np.inf, np.nan][k % 3]
    y[rows9[:5], 41] = np.inf                                 # all five on nonzeros of leaf 9: that cell is summed again
    y[int(ri[cp[3]]), 42] = NA_real
    check(y, "40 light columns + one whose entries all sit on a leaf + an NA")
    y = y0.copy()
    for k in range(60):                                       # 60 x 200 = 12000 listed entries > the list's 8192: all walked
        y[rng.choice(nrow, 200, replace=False), k] = np.inf
    y[rows9[:300], 61] = -np.inf                              # in-between (300 > 256), every entry on a nonzero of leaf 9
    check(y, "list overflow: every dirty column walked")
    check(y0, "clean again")


def test_pbc_fixup_on_two_streams_at_once(hip):
    """The non-finite fix-up is one launch whose steps are separated by a grid-wide barrier (one
    workgroup per CU, kernels_mult_pbc.hip): two products with dirty dense operands in flight on two
    streams must both finish with the one-stream results, and without the barrier's give-up path
    (a second of polling) having been taken."""
    import time
    from sparsearray_amd.device import DeviceCSC, PbcPlan
    dev = torch.device("cuda", 0)
    nrow, ncol, K = 60_000, 1300, 64
    cp, ri, v = random_csc(nrow, ncol, 0.01, seed=77)
    A = _dev(cp, ri, v, nrow)
    rng = np.random.default_rng(78)
    ys = []
    for t in range(2):
        y = rng.uniform(-1, 1, (K, nrow))
        y[3 + t, int(ri[5 + t])] = np.inf             # on a nonzero of leaf 0: exercises the re-summed cells
        y[9, 1234 + t] = np.nan
        ys.append(torch.as_tensor(y, device=dev))
    plans = [PbcPlan(A, K, 40, 16, 7) for _ in range(2)]
    want = []
    for t in range(2):
        out = torch.zeros((K, ncol), dtype=torch.float64, device=dev)
        plans[t].run(ys[t], nrow, out)
        torch.cuda.synchronize()
        want.append(out.clone())
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    outs = [torch.zeros((K, ncol), dtype=torch.float64, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(20):
        for t in range(2):
            with torch.cuda.stream(streams[t]):
                plans[t].run(ys[t], nrow, outs[t])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for t in range(2):
        same_nan = torch.isnan(outs[t]) == torch.isnan(want[t])
        assert bool(same_nan.all())
        fin = torch.isfinite(want[t])
        assert bool((outs[t][fin] == want[t][fin]).all())
    assert dt < 0.5, f"40 small products took {dt:.2f} s: a grid barrier starved"


def test_device_aperm_slab_refused_at_run_time_inside_the_general_form(hip):
    """aperm(x, c(3, 4, 1, 2)) of a 4-d array takes the general form with the slab kernel for its
    first step (aperm(x, c(3, 1, ...)) in one go); one skewed slab over the kernel's cap makes that step refuse at
    run time, INSIDE the part of the workspace behind the general form's intermediates -- it must then go to the key
    sort (which that part is sized for), not re-enter the general form and carve a second set of intermediates."""
    dim, perm = (3000, 4, 6, 5), (3, 4, 1, 2)
    rng = np.random.default_rng(49)
    a = np.zeros(dim, order="F")
    # one slab (fixed indices of axes 2 and 4) holds 12 000 of its 18 000 cells, the other 19 share 80 000
    blk = np.zeros((dim[0], dim[2]))
    blk.reshape(-1)[rng.choice(blk.size, size=12_000, replace=False)] = rng.normal(size=12_000)
    a[:, 0, :, 0] = blk
    mask = np.ones(dim, dtype=bool); mask[:, 0, :, 0] = False
    rest = np.flatnonzero(mask.reshape(-1, order="F"))
    idx = rng.choice(rest, size=80_000, replace=False)
    a.reshape(-1, order="F")[idx] = rng.normal(size=80_000)
    x = SVT_SparseArray.from_dense(a, "double", lacunar=False)
    cp, ri, v = x.to_csc()
    A = _dev(cp, ri, v, dim[0])
    T, new_dim = A.aperm(dim, perm)
    torch.cuda.synchronize()
    want = SVT_SparseArray.from_dense(np.asfortranarray(np.transpose(a, [q - 1 for q in perm])), "double", lacunar=False)
    wcp, wri, wv = want.to_csc()
    assert new_dim == want.dim
    assert np.array_equal(T.col_ptr.cpu().numpy(), wcp)
    assert np.array_equal(T.row_idx.cpu().numpy(), wri)
    assert np.array_equal(T.val.cpu().numpy(), wv)
    # the same array through the 3-d "via" route with a refusing second step: c(3, 2, 1) of (3000, 6, 20)
    a3 = np.asfortranarray(a.transpose(0, 2, 1, 3).reshape((3000, 6, 20), order="F"))
    x3 = SVT_SparseArray.from_dense(a3, "double", lacunar=False)
    cp, ri, v = x3.to_csc()
    T, new_dim = _dev(cp, ri, v, 3000).aperm((3000, 6, 20), (3, 2, 1))
    torch.cuda.synchronize()
    want = SVT_SparseArray.from_dense(np.asfortranarray(np.transpose(a3, (2, 1, 0))), "double", lacunar=False)
    wcp, wri, wv = want.to_csc()
    assert np.array_equal(T.col_ptr.cpu().numpy(), wcp)
    assert np.array_equal(T.row_idx.cpu().numpy(), wri)
    assert np.array_equal(T.val.cpu().numpy(), wv)
