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
