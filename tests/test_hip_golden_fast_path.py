"""The reference's own crossprod / tcrossprod test vectors (tests/golden/golden.json, re-typed from
tests/testthat/test-SparseMatrix-mult.R) through the FAST kernels.

The host entry points send these tiny products to the general gather kernel (the panel kernels need
>= 256 rows and enough work to pay for their layout), so tests/test_hip_golden.py alone never reaches
crossprod_pbc_dma_kernel or its non-finite fix-up.  Here every double case is embedded in a taller
problem that has the same answer -- zero rows appended to the sparse operand, finite zeros appended to
the dense one: the reference's slow path adds 0 * 0 for them (src/SparseVec_dotprod.c:48-65) -- and
run at device level through the panel-blocked layout: the LDS-DMA kernel (40, 16, 7), the gather
kernel (40, 4, 10), and the dense operand given by rows."""
import numpy as np
import pytest
import torch

from helpers import assert_equal, dec, golden_cases
from sparsearray_amd import SVT_SparseArray

pytestmark = pytest.mark.gpu


def _double_product_cases():
    out = []
    for c in golden_cases():
        if c["fn"] not in ("crossprod", "tcrossprod") or "expected" not in c or "error" in c:
            continue
        args = [dec(a, False) for a in c["args"]]
        if any(isinstance(a, SVT_SparseArray) and (a.type != "double" or a.na_background) for a in args):
            continue
        if any(isinstance(a, np.ndarray) and a.dtype != np.float64 for a in args):
            continue
        if not any(isinstance(a, SVT_SparseArray) for a in args):
            continue
        out.append(c)
    return out


CASES = _double_product_cases()


def _dense(a):
    return a.to_dense() if isinstance(a, SVT_SparseArray) else np.asarray(a, dtype=np.float64)


def _csc(m):
    """CSC arrays of a dense matrix whose zeros are the implicit ones (NaN / Inf / NA are stored)."""
    cp, ri, v = [0], [], []
    for j in range(m.shape[1]):
        nz = np.nonzero((m[:, j] != 0) | np.isnan(m[:, j]))[0]
        ri.extend(nz.tolist()); v.extend(m[nz, j].tolist()); cp.append(len(ri))
    return np.asarray(cp, np.int64), np.asarray(ri, np.int32), np.asarray(v, np.float64)


@pytest.mark.parametrize("layout", [(40, 16, 7), (40, 4, 10)], ids=["lds-dma", "gather"])
@pytest.mark.parametrize("case", CASES, ids=[f"{c['id']}-{c['fn']}" for c in CASES])
def test_reference_vectors_through_the_panel_kernels(hip, case, layout):
    from sparsearray_amd.device import DeviceCSC, PbcPlan
    args = [dec(a, False) for a in case["args"]]
    x = args[0]
    y = args[1] if len(args) > 1 else args[0]
    xd, yd = _dense(x), _dense(y)
    if case["fn"] == "tcrossprod":                  # x %*% t(y) = crossprod(t(x), t(y))
        xd, yd = xd.T, yd.T
    exp = dec(case["expected"])
    # the sparse side is whichever operand is an SVT (crossprod(dense, svt) = t(crossprod(svt, dense)))
    flip = not isinstance(x, SVT_SparseArray)
    sp, dn = (yd, xd) if flip else (xd, yd)
    nrow, ncol, K = sp.shape[0], sp.shape[1], dn.shape[1]
    if ncol == 0 or K == 0:
        pytest.skip("zero-extent result: nothing for a kernel to do")
    PAD = 1280 if layout[2] == 10 else 384          # >= one full panel past the data
    spp = np.zeros((PAD, ncol)); spp[:nrow] = sp
    dnp = np.zeros((PAD, K)); dnp[:nrow] = dn
    dev = torch.device("cuda", 0)
    A = DeviceCSC.from_host(PAD, *_csc(spp))
    plan = PbcPlan(A, K, *layout)
    for by_rows in (False, True):
        out = torch.full((K, ncol), 7.0, dtype=torch.float64, device=dev)
        if by_rows:
            plan.run(torch.as_tensor(np.ascontiguousarray(dnp), device=dev), K, out, tr_y=True)
        else:
            plan.run(torch.as_tensor(np.ascontiguousarray(dnp.T), device=dev), PAD, out)
        torch.cuda.synchronize()
        got = out.cpu().numpy().T                   # ncol x K
        if flip:
            got = got.T
        # expect_identical cases (test-SparseMatrix-mult.R:231-246 etc.): the NA-vs-NaN class of every
        # cell must be the reference's, through the panel kernels and their non-finite fix-up too
        assert_equal(got, np.asarray(exp, dtype=np.float64).reshape(got.shape), tol=1e-6,
                     strict_na=case["cmp"] == "identical",
                     what=f"case {case['id']} {case['fn']} [{case['src']}] layout {layout} by_rows {by_rows}")
