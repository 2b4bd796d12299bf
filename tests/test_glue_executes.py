"""integration/svt_hip_glue.c EXECUTED (build container, CPU).

The glue -- the 20 registered `.Call` entry points an R maintainer adds to the reference's src/ -- is compiled
with the reference helper files it calls (read straight from /root/reference/src; the test skips where that
mount is absent, i.e. on the GPU box) against a FUNCTIONAL test-only stand-in for R's C API
(tests/r_api_standin/r_standin.c: fake SEXPs, counted PROTECT stack, guarded R_alloc(), error() / warning()
capture), its svt_* lookups bound to the CPU oracle's ABI (same svt_view layout), and driven through the
registered C_* names with every golden case of tests/golden/ (inputs of the reference's own testthat files):
results, warnings, error strings, names / dimnames of the results, PROTECT depth 0 at exit, no write past an
R_alloc() block, no fall-through to a CPU body.  Interface: src/R_init_SparseArray.c:31-147,
src/SparseMatrix_mult.h:6-43, src/SparseArray_matrixStats.h:6-28, src/rowsum_methods.h:6-36.
What this is NOT: a build of the reference's compute path, or an oracle -- expected values stay those of
tests/golden/golden.json.
"""
import os
import shutil

import numpy as np
import pytest

from helpers import check_case, golden_cases

REF_SRC = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_SRC) or shutil.which("gcc") is None,
                                reason="needs the reference's headers / helper files and gcc (build container only)")

CASES = golden_cases()


@pytest.fixture(scope="module")
def glue(tmp_path_factory):
    import glue_harness
    harness, shim = glue_harness.build(str(tmp_path_factory.mktemp("glue")))
    return glue_harness.Glue(harness, shim)


@pytest.fixture(scope="module")
def glue_session(glue):
    import glue_harness
    from oracle.oracle import oracle_dispatcher
    from sparsearray_amd.api import Session
    return Session(glue_harness.GlueDispatcher(glue, oracle_dispatcher()))


@pytest.mark.parametrize("lacunar", [True, False], ids=["lacunar", "plain"])
@pytest.mark.parametrize("case", CASES, ids=[f"{c['id']}-{c['fn']}" for c in CASES])
def test_golden_case_through_the_glue(glue_session, case, lacunar):
    check_case(glue_session, case, lacunar=lacunar)


def test_every_registered_entry_point_ran(glue, glue_session):
    """After the golden cases: the harness exports the 20 registered names and they have been called."""
    names = ["C_crossprod2_SVT_mat", "C_crossprod2_mat_SVT", "C_crossprod2_SVT_SVT", "C_crossprod1_SVT",
             "C_colStats_SVT", "C_rowStats_SVT", "C_summarize_SVT", "C_rowsum_SVT", "C_colsum_SVT",
             "C_rowsum_dgCMatrix", "C_colsum_dgCMatrix", "C_get_num_procs", "C_get_max_threads", "C_set_max_threads",
             "C_transpose_2D_SVT", "C_aperm_SVT", "C_colMins_dgCMatrix", "C_colMaxs_dgCMatrix",
             "C_colRanges_dgCMatrix", "C_colVars_dgCMatrix"]
    for n in names:
        assert hasattr(glue.lib, n), n
    d = glue_session._call
    prev = d("C_set_max_threads", 2)
    assert d("C_get_max_threads") == 2 and d("C_get_num_procs") >= 1
    assert d("C_set_max_threads", prev) == 2
    assert glue.stats["calls"] > 1000 and glue.stats["max_protect"] >= 2


def test_result_types_of_summarize(glue, glue_session):
    """The R object C_summarize_SVT returns has the type the reference's res2nakedSEXP() gives
    (src/Rvector_summarization.c:1243-1296): integer sums stay integer while they fit, countNAs is an integer,
    any / all / anyNA are logical, range has two values."""
    from sparsearray_amd import SVT_SparseArray
    g = glue
    m = np.zeros((4, 3), dtype=np.int32); m[1, 0] = 7; m[2, 2] = -3
    x = SVT_SparseArray.from_dense(m, type="integer")
    xd, xt, xs = g.svt_args(x)

    def call(op):
        s = g.call("C_summarize_SVT", xd, xt, xs, g.lgl(False), g.string(op), g.lgl(False), g.real(float("nan")))
        return g.lib.sx_type(s), g.to_numpy(s)
    assert call("sum") == (13, pytest.approx(np.array([4]))) and call("sum")[1].dtype == np.int32
    assert call("countNAs")[0] == 13 and call("anyNA")[0] == 10 and call("any")[0] == 10
    t, r = call("range")
    assert t == 13 and r.tolist() == [-3, 7]
    assert call("mean")[0] == 14
    big = SVT_SparseArray.from_dense(np.full((3, 2), 2 ** 30, dtype=np.int32), type="integer")
    bd, bt, bs = g.svt_args(big)
    s = g.call("C_summarize_SVT", bd, bt, bs, g.lgl(False), g.string("sum"), g.lgl(False), g.real(float("nan")))
    assert g.lib.sx_type(s) == 14 and g.to_numpy(s)[0] == 6.0 * 2 ** 30      # past INT_MAX: double
    g.reset()


def test_argument_checks_raise_the_reference_messages(glue):
    """error() paths of the glue itself: the messages of src/SparseMatrix_mult.c:943-966,
    src/SparseArray_matrixStats.c:33-42, src/rowsum_methods.c:15-37, src/SparseArray_aperm.c:455-474."""
    import glue_harness
    from sparsearray_amd import SVT_SparseArray
    g = glue
    x = SVT_SparseArray.from_dense(np.eye(4, 3))
    xd, xt, xs = g.svt_args(x)
    y = g.matrix(np.ones((5, 2)))
    with pytest.raises(glue_harness.RError, match="non-conformable"):
        g.call("C_crossprod2_SVT_mat", xd, xt, xs, y, g.lgl(False), g.string("double"), g.nil)
    with pytest.raises(glue_harness.RError, match="not supported yet"):
        g.call("C_crossprod2_SVT_mat", xd, xt, xs, g.matrix(np.ones((4, 2))), g.lgl(False), g.string("integer"), g.nil)
    with pytest.raises(glue_harness.RError, match="invalid 'x_type' value"):
        g.call("C_crossprod1_SVT", xd, g.string("nonsense"), xs, g.string("double"), g.nil)
    with pytest.raises(glue_harness.RError, match="'dims' must be >= 1 and <= 2"):
        g.call("C_colStats_SVT", xd, g.nil, xt, xs, g.lgl(False), g.string("sum"), g.lgl(False), g.real(0.0), g.ints(3))
    with pytest.raises(glue_harness.RError, match="'na.rm' must be TRUE or FALSE"):
        g.call("C_colStats_SVT", xd, g.nil, xt, xs, g.lgl(False), g.string("sum"), g.ints(1), g.real(0.0), g.ints(1))
    with pytest.raises(glue_harness.RError, match="one element per row"):
        g.call("C_rowsum_SVT", xd, xt, xs, g.ints(1, 2), g.ints(2), g.lgl(False))
    with pytest.raises(glue_harness.RError, match=">= 1 and <= 'ngroup'"):
        g.call("C_rowsum_SVT", xd, xt, xs, g.ints(1, 2, 3, 1), g.ints(2), g.lgl(False))
    with pytest.raises(glue_harness.RError, match="cannot contain duplicates"):
        g.call("C_aperm_SVT", xd, xt, xs, g.ints(1, 1))
    with pytest.raises(glue_harness.RError, match="exactly 2 dimensions"):
        x3 = SVT_SparseArray.from_dense(np.ones((2, 2, 2)))
        g.call("C_transpose_2D_SVT", *g.svt_args(x3))
    g.reset()


def test_null_subtrees_and_dimnames(glue):
    """make_view() on a 3-d tree with NULL subtrees (their leaves stay empty slots), colStats / rowStats results
    carry names / dimnames the way propagate_*_dimnames() does (src/SparseArray_matrixStats.c:108-176)."""
    from sparsearray_amd import SVT_SparseArray
    g = glue
    a = np.zeros((3, 4, 5)); a[1, 2, 0] = 2.5; a[0, 0, 3] = -1.0; a[2, 3, 3] = 4.0      # slabs 1, 2, 4 are NULL subtrees
    dn = [["a", "b", "c"], None, ["s%d" % i for i in range(5)]]
    x = SVT_SparseArray.from_dense(a, dimnames=dn)
    xd, xt, xs = g.svt_args(x)
    s = g.call("C_colStats_SVT", xd, g.dimnames(dn), xt, xs, g.lgl(False), g.string("sum"), g.lgl(False),
               g.real(float("nan")), g.ints(1))
    assert np.array_equal(g.to_numpy(s), a.sum(axis=0)) and g.names_of(s) == [None, dn[2]]
    s = g.call("C_colStats_SVT", xd, g.dimnames(dn), xt, xs, g.lgl(False), g.string("sum"), g.lgl(False),
               g.real(float("nan")), g.ints(2))
    assert np.array_equal(g.to_numpy(s), a.sum(axis=(0, 1))) and g.names_of(s) == [dn[2]]
    s = g.call("C_rowStats_SVT", xd, g.dimnames(dn), xt, xs, g.lgl(False), g.string("sum"), g.lgl(False), g.nil, g.ints(2))
    assert np.array_equal(g.to_numpy(s), a.sum(axis=2)) and g.names_of(s) == [dn[0], None]
    s = g.call("C_rowStats_SVT", xd, g.dimnames(dn), xt, xs, g.lgl(False), g.string("sum"), g.lgl(False), g.nil, g.ints(1))
    assert np.array_equal(g.to_numpy(s), a.sum(axis=(1, 2))) and g.names_of(s) == [dn[0]]
    g.reset()


def test_status_above_zero_runs_the_cpu_body_and_below_zero_raises(glue):
    """include/svt_hip.h: a status > 0 means "not supported here" (e.g. 2^31 nonzeros or more in a transposition, an
    operation the device kernels do not implement) -- every registered compute entry point must then hand the call
    to the reference's own body (its `_cpu` name) with nothing left protected; a status < 0 is error() with the
    library's message.  The shim answers every svt_* call with $SVT_SHIM_STATUS here."""
    from sparsearray_amd import SVT_SparseArray
    g = glue
    a = np.zeros((6, 4)); a[1, 0] = 2.5; a[3, 2] = -1.0; a[5, 3] = 4.0
    x = SVT_SparseArray.from_dense(a)
    a3 = np.zeros((3, 4, 2)); a3[1, 2, 0] = 2.5; a3[0, 0, 1] = -1.0
    x3 = SVT_SparseArray.from_dense(a3)
    y = np.arange(12, dtype=np.float64).reshape(6, 2, order="F")

    def dgc():
        return g.dgc(((6, 4), np.array([0, 1, 1, 2, 3], dtype=np.int32), np.array([1, 3, 5], dtype=np.int32),
                      np.array([2.5, -1.0, 4.0])))

    NAMES = ["C_crossprod2_SVT_mat", "C_crossprod2_mat_SVT", "C_crossprod2_SVT_SVT", "C_crossprod1_SVT", "C_colStats_SVT",
             "C_rowStats_SVT", "C_summarize_SVT", "C_rowsum_SVT", "C_colsum_SVT", "C_rowsum_dgCMatrix", "C_colsum_dgCMatrix",
             "C_colMins_dgCMatrix", "C_colMaxs_dgCMatrix", "C_colRanges_dgCMatrix", "C_colVars_dgCMatrix",
             "C_transpose_2D_SVT", "C_aperm_SVT"]

    def args_of(name):          # (fresh SEXPs per call: g.reset() releases them all)
        xd, xt, xs = g.svt_args(x)
        xd3, xt3, xs3 = g.svt_args(x3)
        grp = g.ints(1, 2, 1, 2, 1, 2)
        grp_c = g.ints(1, 2, 1, 2)
        F, nan = g.lgl(False), g.real(float("nan"))
        return {
            "C_crossprod2_SVT_mat": (xd, xt, xs, g.matrix(y), F, g.string("double"), g.nil),
            "C_crossprod2_mat_SVT": (g.matrix(y), xd, xt, xs, F, g.string("double"), g.nil),
            "C_crossprod2_SVT_SVT": (xd, xt, xs, xd, xt, xs, g.string("double"), g.nil),
            "C_crossprod1_SVT": (xd, xt, xs, g.string("double"), g.nil),
            "C_colStats_SVT": (xd, g.nil, xt, xs, F, g.string("sum"), F, nan, g.ints(1)),
            "C_rowStats_SVT": (xd3, g.nil, xt3, xs3, F, g.string("sum"), F, g.nil, g.ints(1)),
            "C_summarize_SVT": (xd, xt, xs, F, g.string("sum"), F, nan),
            "C_rowsum_SVT": (xd, xt, xs, grp, g.ints(2), F),
            "C_colsum_SVT": (xd, xt, xs, grp_c, g.ints(2), F),
            "C_rowsum_dgCMatrix": (dgc(), grp, g.ints(2), F),
            "C_colsum_dgCMatrix": (dgc(), grp_c, g.ints(2), F),
            "C_colMins_dgCMatrix": (dgc(), F), "C_colMaxs_dgCMatrix": (dgc(), F),
            "C_colRanges_dgCMatrix": (dgc(), F), "C_colVars_dgCMatrix": (dgc(), F),
            "C_transpose_2D_SVT": (xd, xt, xs),
            "C_aperm_SVT": (xd3, xt3, xs3, g.ints(2, 1, 3)),
        }[name]
    import glue_harness
    try:
        os.environ["SVT_SHIM_STATUS"] = "1"
        for name in NAMES:
            g.call(name, *args_of(name), allow_cpu_body=True)        # (asserts protection depth 0 and intact R_alloc() guards)
            assert g.lib.env_cpu_body_calls() == 1, name
            assert g.lib.env_cpu_body_last().decode() == name + "_cpu", (name, g.lib.env_cpu_body_last())
            g.reset()
        os.environ["SVT_SHIM_STATUS"] = "-1"
        for name in NAMES:
            with pytest.raises(glue_harness.RError, match="forced status"):
                g.call(name, *args_of(name))
            assert g.lib.env_cpu_body_calls() == 0, name
            g.reset()
    finally:
        os.environ.pop("SVT_SHIM_STATUS", None)
