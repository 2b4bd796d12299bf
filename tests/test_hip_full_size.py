"""BASELINE.json config 2a at full size (1e6 x 1e4 @ 1 %, dense 1e6 x 128) on the GPU:
the oracle cannot finish these in seconds, so parity is checked through
size-independent properties and independent device code paths."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NROW, NCOL, DENS, K = 1_000_000, 10_000, 0.01, 128


@pytest.fixture(scope="module")
def operands(hip):
    from sparsearray_amd import synth
    from sparsearray_amd.device import DeviceCSC, PbcPlan
    dev = torch.device("cuda", 0)
    cp, ri, v = synth.random_device_csc(NROW, NCOL, DENS, seed=7, device=dev)
    A = DeviceCSC(NROW, cp, ri, v)
    plan = PbcPlan(A, K)
    Y = synth.random_dense(NROW, K, seed=107, device=dev)          # (K, nrow) == col-major nrow x K
    return A, plan, Y


def _crossprod(plan, Y):
    out = torch.zeros((K, NCOL), dtype=torch.float64, device=Y.device)
    plan.run(Y, NROW, out)
    torch.cuda.synchronize()
    return out


def test_full_size_matches_direct_gather_on_sampled_columns(operands):
    """64 random leaves recomputed with plain torch gathers (ascending offsets)."""
    A, plan, Y = operands
    out = _crossprod(plan, Y)
    cols = torch.randint(0, NCOL, (64,), generator=torch.Generator().manual_seed(1)).tolist()
    worst = 0.0
    for c in cols:
        lo, hi = int(A.col_ptr[c]), int(A.col_ptr[c + 1])
        rows = A.row_idx[lo:hi].long()
        want = (Y[:, rows] * A.val[lo:hi]).sum(dim=1)                # (K,)
        scale = (Y[:, rows] * A.val[lo:hi]).abs().sum(dim=1).clamp_min(1e-300)
        worst = max(worst, float(((out[:, c] - want).abs() / scale).max()))
    assert worst <= 1e-12, worst


def test_full_size_column_sums_identity(operands):
    """sum_c crossprod(A, Y)[c, k] == sum_r rowSums(A)[r] * Y[r, k] -- the right-hand
    side comes from the row-stats kernel, an independent code path."""
    from sparsearray_amd.device import rowsums
    A, plan, Y = operands
    out = _crossprod(plan, Y)
    rs = rowsums(A)
    torch.cuda.synchronize()
    lhs = out.sum(dim=1)
    rhs = (Y * rs).sum(dim=1)
    scale = (Y.abs() * rs.abs()).sum(dim=1)
    assert float(((lhs - rhs).abs() / scale).max()) <= 1e-11


def test_full_size_ones_give_colsums_and_linearity(operands):
    from sparsearray_amd.device import colstats
    A, plan, Y = operands
    ones = torch.ones_like(Y)
    cs, _ = colstats(A, "sum")
    out1 = _crossprod(plan, ones)
    err = (out1 - cs[None, :]).abs().max() / cs.abs().max()
    assert float(err) <= 1e-12
    # linearity in the dense operand
    Y2 = torch.flip(Y, dims=[0]) * 0.5
    o_a = _crossprod(plan, Y)
    o_b = _crossprod(plan, Y2)
    o_ab = _crossprod(plan, Y + 2.0 * Y2)
    scale = o_a.abs().max() + 2 * o_b.abs().max()
    assert float((o_ab - (o_a + 2.0 * o_b)).abs().max() / scale) <= 1e-12


def test_full_size_poisoned_dense_entry_switches_semantics(operands):
    """One Inf in Y: every leaf gets NaN/Inf in that dense column (0 * Inf at the
    implicit zeros, src/SparseVec_dotprod.c:48-65), other dense columns unchanged."""
    A, plan, Y = operands
    clean = _crossprod(plan, Y)
    Yp = Y.clone()
    Yp[5, 123_457] = float("inf")
    got = _crossprod(plan, Yp)
    assert not torch.isfinite(got[5]).any()
    keep = [k for k in range(K) if k != 5]
    assert float((got[keep] - clean[keep]).abs().max()) <= 1e-9 * float(clean.abs().max())


def test_full_size_dense_operand_given_by_rows(operands):
    """tcrossprod orientation at full size: the 128 x 1e6 operand is transposed on the device and
    takes the product kernel; results identical to the column-major call."""
    A, plan, Y = operands
    want = _crossprod(plan, Y)
    Yrm = Y.t().contiguous()                                  # (nrow, K) C-contiguous = column-major K x nrow
    out = torch.zeros((K, NCOL), dtype=torch.float64, device=Y.device)
    plan.run(Yrm, K, out, tr_y=True)
    torch.cuda.synchronize()
    assert torch.equal(out, want)


# ---------------------------------------------------------------------------
# BASELINE.json config 5: 2e4 x 2e4 x 64 @ 0.5 % (1.28e8 nonzeros, 1.28e6 leaves)
# ---------------------------------------------------------------------------
D5 = (20_000, 20_000, 64)


@pytest.fixture(scope="module")
def array5(hip):
    from sparsearray_amd import synth
    from sparsearray_amd.device import DeviceCSC
    dev = torch.device("cuda", 0)
    cp, ri, v = synth.random_device_csc(D5[0], D5[1] * D5[2], 0.005, seed=5, device=dev)
    return DeviceCSC(D5[0], cp, ri, v)


def test_config5_col_stats_match_segment_sums(array5):
    """colSums(dims=1) (one result per leaf) and colSums(dims=2) (one per 2e4 leaves)
    against prefix sums of the value array (torch)."""
    from sparsearray_amd.device import colstats
    A = array5
    csum = torch.cumsum(A.val, 0)
    csum = torch.cat([torch.zeros(1, dtype=csum.dtype, device=csum.device), csum])
    want1 = csum[A.col_ptr[1:]] - csum[A.col_ptr[:-1]]
    got1, _ = colstats(A, "sum")
    torch.cuda.synchronize()
    scale = torch.cumsum(A.val.abs(), 0)[-1] / A.ncol * 50 + 1.0
    assert float((got1 - want1).abs().max()) <= 1e-9 * float(scale)
    got2, _ = colstats(A, "sum", inner=D5[1])
    torch.cuda.synchronize()
    want2 = want1.view(D5[2], D5[1]).sum(dim=1)
    assert got2.numel() == D5[2]
    assert float(((got2 - want2).abs() / want2.abs().clamp_min(1.0)).max()) <= 1e-9
    # colMeans / colVars agree with each other through the textbook identity
    # on a sample of leaves
    mean, _ = colstats(A, "mean")
    var, _ = colstats(A, "var1")
    torch.cuda.synchronize()
    for j in (0, 12345, A.ncol - 1):
        lo, hi = int(A.col_ptr[j]), int(A.col_ptr[j + 1])
        x = torch.zeros(D5[0], dtype=torch.float64, device=A.val.device)
        x[A.row_idx[lo:hi].long()] = A.val[lo:hi]
        assert abs(float(mean[j]) - float(x.mean())) <= 1e-12
        assert abs(float(var[j]) - float(x.var(unbiased=True))) <= 1e-12 * max(1.0, float(x.var()))


def test_config5_rowsums_dims2_match_scatter_add(array5):
    """rowSums(dims=2): out[r, j2] = sum over the 64 slices -- 2e4 x 2e4 doubles (3.2 GB) --
    against torch.index_add_ on the flattened output."""
    from sparsearray_amd.device import rowsums
    A = array5
    inner = D5[1]
    got = rowsums(A, inner=inner)
    torch.cuda.synchronize()
    leaf = torch.repeat_interleave(torch.arange(A.ncol, device=A.val.device), A.col_ptr[1:] - A.col_ptr[:-1])
    cell = (leaf % inner) * D5[0] + A.row_idx.long()
    del leaf
    want = torch.zeros(inner * D5[0], dtype=torch.float64, device=A.val.device)
    want.index_add_(0, cell, A.val)
    torch.cuda.synchronize()
    assert float((got - want).abs().max()) <= 1e-11
    assert float(got.abs().sum()) > 0


# ---------------------------------------------------------------------------
# BASELINE config 4 at full size on ONE GPU (1e7 x 5e4 @ 0.1 %, 5e8 nonzeros, 6 GB of CSC):
# the record stream is > 4 GiB, the kernels' 32-bit cursors are per column group.
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("Kc", [64, 128], ids=["K64-gather", "K128-gather2"])
def test_config4_full_size_crossprod_and_colsums(hip, Kc):
    """Kc = 64: crossprod_pbc_gather_kernel (lane = dense column); Kc = 128 (BASELINE's width): the kernel with
    two dense columns per lane."""
    from sparsearray_amd import synth
    from sparsearray_amd.device import DeviceCSC, PbcPlan, colstats
    dev = torch.device("cuda", 0)
    nrow, ncol = 10_000_000, 50_000
    cp, ri, v = synth.random_device_csc(nrow, ncol, 0.001, seed=4, device=dev)
    A = DeviceCSC(nrow, cp, ri, v)
    assert A.nnz > 4.9e8
    plan = PbcPlan(A, Kc, 0, 0, 0)                  # layout by density: the gather kernel at 0.1 %
    Y = synth.random_dense(nrow, Kc, seed=104, device=dev)
    out = torch.zeros((Kc, ncol), dtype=torch.float64, device=dev)
    plan.run(Y, nrow, out)
    torch.cuda.synchronize()
    g = torch.Generator().manual_seed(2)
    cols = torch.randint(0, ncol, (48,), generator=g).tolist() + [0, ncol - 1, ncol // 2]
    worst = 0.0
    for c in cols:
        lo, hi = int(A.col_ptr[c]), int(A.col_ptr[c + 1])
        rows = A.row_idx[lo:hi].long()
        prod = Y[:, rows] * A.val[lo:hi]
        worst = max(worst, float(((out[:, c] - prod.sum(dim=1)).abs() /
                                  prod.abs().sum(dim=1).clamp_min(1e-300)).max()))
    assert worst <= 1e-12, worst
    # dense operand of ones: every dense column of the product is colSums(A)
    cs, _ = colstats(A, "sum")
    Y.fill_(1.0)
    plan.run(Y, nrow, out)
    torch.cuda.synchronize()
    assert float((out - cs[None, :]).abs().max() / cs.abs().max()) <= 1e-12
    # colSums against a segmented torch reduction
    want = torch.zeros(ncol, dtype=torch.float64, device=dev)
    seg = torch.repeat_interleave(torch.arange(ncol, device=dev), A.col_ptr[1:] - A.col_ptr[:-1])
    want.index_add_(0, seg, A.val)
    assert float((cs - want).abs().max() / want.abs().max()) <= 1e-12
