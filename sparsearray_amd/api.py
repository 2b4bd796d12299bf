"""R-level operator surface of the hot path, mirrored in Python.

Each function restates the argument checking / dispatch of the reference's
S4 method and then goes through ``SparseArray_Call(.NAME, ...)`` -- exactly
one ``.Call`` entry point per operation, the same names and argument meaning
as ``src/R_init_SparseArray.c:94,121-134``:

    C_crossprod2_SVT_mat  C_crossprod2_mat_SVT  C_crossprod2_SVT_SVT
    C_crossprod1_SVT      C_colStats_SVT        C_rowStats_SVT
    C_summarize_SVT       C_rowsum_SVT          C_colsum_SVT
    C_rowsum_dgCMatrix    C_colsum_dgCMatrix
    C_colMins_dgCMatrix   C_colMaxs_dgCMatrix   C_colRanges_dgCMatrix  C_colVars_dgCMatrix
    C_transpose_2D_SVT    C_aperm_SVT

The product binds those names to the HIP library (``sparsearray_amd._hip``);
there is no CPU implementation in this package.  A ``Session`` can be built
around any other dispatcher with the same entry points -- the test-suite does
that with the CPU oracle so both run through identical R-level logic.

Conventions: dense inputs/outputs are numpy arrays with R index semantics;
"logical" results are int32 with ``NA_integer`` for NA.
"""
from __future__ import annotations

import warnings
from typing import Callable, Optional

import numpy as np

from .svt import (NA_integer, NA_real, SVT_SparseArray, is_NA_real, r_type_of)


class SparseArrayError(RuntimeError):
    """R's error() raised by an entry point."""


class SparseArrayUnsupported(SparseArrayError):
    """Status > 0 of the C ABI (include/svt_hip.h): the device kernels do not take this operand / operation; the R
    glue runs the reference's CPU body instead (integration/svt_hip_glue.c).  This package has no CPU path: it
    raises."""


_SUPPORTED_MULT_TYPES = ("double", "integer")


def _check_crossprod_input_type(type_: str):
    # R/SparseMatrix-mult.R:13-20
    if type_ not in _SUPPORTED_MULT_TYPES:
        raise SparseArrayError(
            "input objects must be of type() \"double\" or \"integer\"")


def _common_type(t1: str, t2: str) -> str:
    order = {"logical": 0, "integer": 1, "double": 2}
    return t1 if order[t1] >= order[t2] else t2


def _as_R_matrix(y) -> np.ndarray:
    y = np.asarray(y)
    if y.ndim != 2:
        raise SparseArrayError("input objects must have 2 dimensions")
    if y.dtype == np.bool_:
        y = y.astype(np.int32)
    if y.dtype not in (np.float64, np.int32):
        raise TypeError("dense operands must be float64 or int32")
    return y


def _dense_to_double(y: np.ndarray) -> np.ndarray:
    if y.dtype == np.float64:
        return y
    out = y.astype(np.float64)
    out[y == NA_integer] = NA_real
    return out


class Session:
    """All R-level generics of the hot path over one ``.Call`` dispatcher."""

    def __init__(self, call: Callable):
        self._call = call

    # Resident operands: keep device copies of SVT operands across calls (opt-in; see
    # include/svt_hip.h).  In the R package this would be an option() read by the glue.
    def resident_set_limit(self, nbytes: int):
        self._call.resident_set_limit(nbytes)

    def resident_clear(self):
        self._call.resident_clear()

    def resident_stats(self) -> dict:
        return self._call.resident_stats()

    # SparseArray.Call(), R/thread-control.R:87-92
    def SparseArray_Call(self, name: str, *args):
        return self._call(name, *args)

    # ------------------------------------------------------------------
    # crossprod / tcrossprod / %*%   (R/SparseMatrix-mult.R)
    # ------------------------------------------------------------------
    def _crossprod2_SparseMatrix_matrix(self, x, y, transpose_y=False):
        y = _as_R_matrix(y)
        if x.ndim != 2:
            raise SparseArrayError("input objects must have 2 dimensions")
        if transpose_y:
            if x.dim[0] != y.shape[1]:
                raise SparseArrayError("non-conformable arguments")
        elif x.dim[0] != y.shape[0]:
            raise SparseArrayError("non-conformable arguments")
        ytype = r_type_of(y)
        if x.type == ytype:
            _check_crossprod_input_type(x.type)
        else:
            xy = _common_type(x.type, ytype)
            _check_crossprod_input_type(xy)
            if not self._device_coerces(x.type, ytype):
                x = x.with_type(xy)
                y = _dense_to_double(y)
        return self.SparseArray_Call("C_crossprod2_SVT_mat", x, y,
                                     bool(transpose_y))

    def _crossprod2_matrix_SparseMatrix(self, x, y, transpose_x=False):
        x = _as_R_matrix(x)
        if y.ndim != 2:
            raise SparseArrayError("input objects must have 2 dimensions")
        if transpose_x:
            if x.shape[1] != y.dim[0]:
                raise SparseArrayError("non-conformable arguments")
        elif x.shape[0] != y.dim[0]:
            raise SparseArrayError("non-conformable arguments")
        xtype = r_type_of(x)
        if xtype == y.type:
            _check_crossprod_input_type(y.type)
        else:
            xy = _common_type(xtype, y.type)
            _check_crossprod_input_type(xy)
            if not self._device_coerces(xtype, y.type):
                y = y.with_type(xy)
                x = _dense_to_double(x)
        return self.SparseArray_Call("C_crossprod2_mat_SVT", x, y,
                                     bool(transpose_x))

    def _device_coerces(self, t1, t2) -> bool:
        # integer x double: the R methods coerce the integer operand first (type(x) <- "double");
        # the HIP library takes the pair as it is and widens on the device
        return getattr(self._call, "accepts_mixed_types", False) and \
            {t1, t2} == {"integer", "double"}

    def _crossprod2_SparseMatrix_SparseMatrix(self, x, y):
        if x.ndim != 2 or y.ndim != 2:
            raise SparseArrayError("input objects must have 2 dimensions")
        if x.dim[0] != y.dim[0]:
            raise SparseArrayError("non-conformable arguments")
        if x.type == y.type:
            _check_crossprod_input_type(x.type)
        else:
            xy = _common_type(x.type, y.type)
            _check_crossprod_input_type(xy)
            x, y = x.with_type(xy), y.with_type(xy)
        return self.SparseArray_Call("C_crossprod2_SVT_SVT", x, y)

    def _matmul_fused(self, x, y):
        # argument checks of .crossprod2_SparseMatrix_{matrix,SparseMatrix} on (t(x), y)
        if x.ndim != 2:
            raise SparseArrayError("input objects must have 2 dimensions")
        if isinstance(y, SVT_SparseArray):
            if y.ndim != 2:
                raise SparseArrayError("input objects must have 2 dimensions")
            if x.dim[1] != y.dim[0]:
                raise SparseArrayError("non-conformable arguments")
            ytype = y.type
        else:
            y = _as_R_matrix(y)
            if x.dim[1] != y.shape[0]:
                raise SparseArrayError("non-conformable arguments")
            ytype = r_type_of(y)
        if x.type == ytype:
            _check_crossprod_input_type(x.type)
        else:
            xy = _common_type(x.type, ytype)
            _check_crossprod_input_type(xy)
            if isinstance(y, SVT_SparseArray) or not self._device_coerces(x.type, ytype):
                x = x.with_type(xy)
                y = y.with_type(xy) if isinstance(y, SVT_SparseArray) else _dense_to_double(y)
        if isinstance(y, SVT_SparseArray):
            return self.SparseArray_Call("C_matmul_SVT_SVT", x, y)
        return self.SparseArray_Call("C_matmul_SVT_mat", x, y)

    def _crossprod1_SparseMatrix(self, x):
        if x.ndim != 2:
            raise SparseArrayError("'x' must have 2 dimensions")
        _check_crossprod_input_type(x.type)
        return self.SparseArray_Call("C_crossprod1_SVT", x)

    @staticmethod
    def _no_NaArray(what, *objs):
        """crossprod()/%*%/rowsum() have methods for SVT_SparseMatrix only
        (R/SparseMatrix-mult.R, R/rowsum-methods.R): an NaArray operand is an error."""
        for o in objs:
            if isinstance(o, SVT_SparseArray) and o.na_background:
                raise SparseArrayError(f"unable to find an inherited method for function "
                                       f"'{what}' for signature 'x = \"NaMatrix\"'")

    def t(self, x):
        """t(x) of an SVT_SparseMatrix: t.SVT_SparseMatrix, R/SparseArray-aperm.R:11-20 =
        one C_transpose_2D_SVT call (src/SparseArray_aperm.c:395-423)."""
        if x.ndim != 2:
            raise SparseArrayError("object to transpose must have exactly 2 dimensions")
        return self.SparseArray_Call("C_transpose_2D_SVT", x)

    def crossprod(self, x, y=None):
        self._no_NaArray("crossprod", x, y)
        xs, ys = isinstance(x, SVT_SparseArray), isinstance(y, SVT_SparseArray)
        if xs and y is None:
            return self._crossprod1_SparseMatrix(x)
        if xs and ys:
            return self._crossprod2_SparseMatrix_SparseMatrix(x, y)
        if xs:
            return self._crossprod2_SparseMatrix_matrix(x, y)
        if ys:
            return self._crossprod2_matrix_SparseMatrix(x, y)
        raise TypeError("crossprod() needs at least one SVT_SparseArray")

    def tcrossprod(self, x, y=None):
        self._no_NaArray("tcrossprod", x, y)
        xs, ys = isinstance(x, SVT_SparseArray), isinstance(y, SVT_SparseArray)
        # The R methods transpose first (t() = C_transpose_2D_SVT on the host, R/SparseMatrix-mult.R:165-193).  The HIP
        # library offers both sparse forms in one call with the transpositions on the device (svt_tcrossprod*_SVT*,
        # include/svt_hip.h); same checks, same coercions, applied to the transposed operands.
        has = getattr(self._call, "has_entry", lambda name: False)
        if xs and y is None:
            if has("C_tcrossprod1_SVT") and x.ndim == 2:
                _check_crossprod_input_type(x.type)
                return self.SparseArray_Call("C_tcrossprod1_SVT", x)
            return self._crossprod1_SparseMatrix(self.t(x))
        if xs and ys:
            if has("C_tcrossprod2_SVT_SVT") and x.ndim == 2 and y.ndim == 2:
                if x.dim[1] != y.dim[1]:
                    raise SparseArrayError("non-conformable arguments")
                if x.type == y.type:
                    _check_crossprod_input_type(x.type)
                else:
                    xy = _common_type(x.type, y.type)
                    _check_crossprod_input_type(xy)
                    x, y = x.with_type(xy), y.with_type(xy)
                return self.SparseArray_Call("C_tcrossprod2_SVT_SVT", x, y)
            return self._crossprod2_SparseMatrix_SparseMatrix(self.t(x), self.t(y))
        if xs:
            return self._crossprod2_SparseMatrix_matrix(self.t(x), y, True)
        if ys:
            return self._crossprod2_matrix_SparseMatrix(x, self.t(y), True)
        raise TypeError("tcrossprod() needs at least one SVT_SparseArray")

    def matmul(self, x, y):
        """``x %*% y`` (R/SparseMatrix-mult.R:195-215)."""
        self._no_NaArray("%*%", x, y)
        xs, ys = isinstance(x, SVT_SparseArray), isinstance(y, SVT_SparseArray)
        # The R methods transpose x first (t() = C_transpose_2D_SVT on the host).  The
        # HIP library offers the product in one call, with the transposition on the
        # device (svt_matmul_SVT_*, include/svt_hip.h); same checks, same coercions.
        has = getattr(self._call, "has_entry", lambda name: False)
        if xs and ys:
            if has("C_matmul_SVT_SVT"):
                return self._matmul_fused(x, y)
            return self._crossprod2_SparseMatrix_SparseMatrix(self.t(x), y)
        if xs:
            if has("C_matmul_SVT_mat"):
                return self._matmul_fused(x, y)
            return self._crossprod2_SparseMatrix_matrix(self.t(x), y)
        if ys:
            return self._crossprod2_matrix_SparseMatrix(x, y, True)
        raise TypeError("%*% needs at least one SVT_SparseArray")

    # ------------------------------------------------------------------
    # matrixStats  (R/SparseArray-matrixStats.R)
    # ------------------------------------------------------------------
    def _colStats(self, op, x, na_rm=False, center=None, dims=1):
        # .colStats_SparseArray, R/SparseArray-matrixStats.R:68-107
        dims = int(dims)
        if dims <= 0 or dims > x.ndim:
            raise SparseArrayError(
                "'dims' must be a single integer that is > 0 and <= "
                "length(dim(x)) for the col*() functions, and >= 0 and < "
                "length(dim(x)) for the row*() functions")
        if not isinstance(na_rm, (bool, np.bool_)):
            raise SparseArrayError("'na.rm' must be TRUE or FALSE")
        center = NA_real if center is None else float(center)
        ans, warn = self.SparseArray_Call("C_colStats_SVT", x, op,
                                          bool(na_rm), center, dims)
        if warn:
            warnings.warn("NAs introduced by coercion of "
                          "infinite values to integers")
        return ans

    def colMedians(self, x, na_rm=False):
        """colMedians(x, na.rm) (R/SparseArray-matrixStats.R:786-800; 2-D objects only)."""
        if x.ndim != 2:
            raise SparseArrayError(
                "the colMedians() method for SparseArray objects only supports 2D "
                "objects (i.e. SparseMatrix objects) at the moment")
        if not isinstance(na_rm, (bool, np.bool_)):
            raise SparseArrayError("'na.rm' must be TRUE or FALSE")
        if x.dim[0] == 0:
            return np.full(x.dim[1], NA_real)            # :771-772
        return self.SparseArray_Call("C_colMedians_SVT", x, bool(na_rm))

    def rowMedians(self, x, na_rm=False):
        """rowMedians(x) = colMedians(t(x)) (R/SparseArray-matrixStats.R:802-815)."""
        if x.ndim != 2:
            raise SparseArrayError(
                "the rowMedians() method for SparseArray objects only supports 2D "
                "objects (i.e. SparseMatrix objects) at the moment")
        if not isinstance(na_rm, (bool, np.bool_)):
            raise SparseArrayError("'na.rm' must be TRUE or FALSE")
        has = getattr(self._call, "has_entry", lambda name: False)
        if has("C_rowMedians_SVT") and x.dim[1] > 0 and x.dim[0] > 0:
            return self.SparseArray_Call("C_rowMedians_SVT", x, bool(na_rm))   # t(x) on the device
        return self.colMedians(self.t(x), na_rm=na_rm)

    def _rowStats(self, op, x, na_rm=False, center=None, dims=1):
        # .rowStats_SparseArray, R/SparseArray-matrixStats.R:197-259
        dims = int(dims)
        if dims < 0 or dims >= x.ndim:
            raise SparseArrayError(
                "'dims' must be a single integer that is > 0 and <= "
                "length(dim(x)) for the col*() functions, and >= 0 and < "
                "length(dim(x)) for the row*() functions")
        if dims == 0:
            return self._colStats(op, x, na_rm, center, x.ndim)
        if x.na_background and op not in ("countNAs", "anyNA", "min", "max", "sum"):
            # rowAnys/Alls/Prods/Means/Vars/Sds: no NaArray methods (commented out in
            # R/NaArray-matrixStats.R:187-330)
            raise SparseArrayError(f"unable to find an inherited method for the row {op} "
                                   f"statistic for signature 'x = \"NaArray\"'")
        if op not in ("countNAs", "anyNA", "min", "max", "sum",
                      "centered_X2_sum"):
            return self._OLD_rowStats(op, x, na_rm, center, dims)
        if center is not None:
            ans_dim = x.dim[:dims]
            center = np.asarray(center, dtype=np.float64)
            n = int(np.prod(ans_dim))
            if center.ndim >= 1 and center.shape == tuple(ans_dim):
                pass
            elif center.size in (1, n):
                center = np.broadcast_to(center.reshape(-1, order="F"), (n,)) \
                    if center.size == 1 else center
                center = np.reshape(center, ans_dim, order="F")
            else:
                raise SparseArrayError("unexpected 'center' length")
        ans, warn = self.SparseArray_Call("C_rowStats_SVT", x, op,
                                          bool(na_rm), center, dims)
        if warn:
            warnings.warn("NAs introduced by coercion of "
                          "infinite values to integers")
        return ans

    def aperm(self, x, perm=None):
        """aperm(x, perm) (R/SparseArray-aperm.R:24-60); perm is 1-based, default: reversal."""
        if perm is None:
            perm = list(range(x.ndim, 0, -1))
        perm = [int(p) for p in perm]
        if len(perm) != x.ndim or sorted(perm) != list(range(1, x.ndim + 1)):
            raise SparseArrayError(f"'perm' must be a permutation of 1:{x.ndim}")
        if perm == list(range(1, x.ndim + 1)):
            return x
        return self.SparseArray_Call("C_aperm_SVT", x, perm)

    def _OLD_rowStats(self, op, x, na_rm, center, dims):
        # .OLD_rowStats_SparseArray (:122-190): "aperm(colStats(aperm(x), dims=ndim-dims))",
        # the semantically plain form of :115-118 (the slice-wise tricks of :150-189
        # only avoid the reference's expensive multidimensional transposition)
        tx = self.t(x) if x.ndim == 2 else self.aperm(x)
        ans = self._colStats(op, tx, na_rm, center, x.ndim - dims)
        if isinstance(ans, np.ndarray) and ans.ndim > 1:
            ans = np.ascontiguousarray(np.transpose(ans))
            ans = np.asfortranarray(ans)
        return ans

    def _colCountVals(self, x, na_rm=False, dims=1):
        ans = float(np.prod(x.dim[:dims], dtype=np.float64))
        if na_rm:
            ans = ans - self._colStats("countNAs", x, dims=dims)
        return ans

    def _rowCountVals(self, x, na_rm=False, dims=1):
        ans = float(np.prod(x.dim[dims:], dtype=np.float64))
        if na_rm:
            ans = ans - self._rowStats("countNAs", x, dims=dims)
        return ans

    def colAnyNAs(self, x, dims=1): return self._colStats("anyNA", x, dims=dims)
    def rowAnyNAs(self, x, dims=1): return self._rowStats("anyNA", x, dims=dims)
    def colCountNAs(self, x, dims=1): return self._colStats("countNAs", x, dims=dims)
    def rowCountNAs(self, x, dims=1): return self._rowStats("countNAs", x, dims=dims)
    def colAnys(self, x, na_rm=False, dims=1): return self._colStats("any", x, na_rm, dims=dims)
    def rowAnys(self, x, na_rm=False, dims=1): return self._rowStats("any", x, na_rm, dims=dims)
    def colAlls(self, x, na_rm=False, dims=1): return self._colStats("all", x, na_rm, dims=dims)
    def rowAlls(self, x, na_rm=False, dims=1): return self._rowStats("all", x, na_rm, dims=dims)
    def colMins(self, x, na_rm=False, dims=1): return self._colStats("min", x, na_rm, dims=dims)
    def rowMins(self, x, na_rm=False, dims=1): return self._rowStats("min", x, na_rm, dims=dims)
    def colMaxs(self, x, na_rm=False, dims=1): return self._colStats("max", x, na_rm, dims=dims)
    def rowMaxs(self, x, na_rm=False, dims=1): return self._rowStats("max", x, na_rm, dims=dims)

    def colRanges(self, x, na_rm=False, dims=1):
        mins = self.colMins(x, na_rm, dims)
        maxs = self.colMaxs(x, na_rm, dims)
        return np.stack([mins, maxs], axis=-1)

    def rowRanges(self, x, na_rm=False, dims=1):
        mins = self.rowMins(x, na_rm, dims)
        maxs = self.rowMaxs(x, na_rm, dims)
        return np.stack([mins, maxs], axis=-1)

    def colSums(self, x, na_rm=False, dims=1): return self._colStats("sum", x, na_rm, dims=dims)
    def rowSums(self, x, na_rm=False, dims=1): return self._rowStats("sum", x, na_rm, dims=dims)
    def colProds(self, x, na_rm=False, dims=1): return self._colStats("prod", x, na_rm, dims=dims)
    def rowProds(self, x, na_rm=False, dims=1): return self._rowStats("prod", x, na_rm, dims=dims)
    def colMeans(self, x, na_rm=False, dims=1): return self._colStats("mean", x, na_rm, dims=dims)

    def rowMeans(self, x, na_rm=False, dims=1):
        # :511-516
        if x.na_background:
            return self._rowStats("mean", x, na_rm, dims=dims)     # raises: no NaArray method
        sums = self.rowSums(x, na_rm, dims)
        nvals = self._rowCountVals(x, na_rm, dims)
        with np.errstate(all="ignore"):
            return sums / nvals

    colSums2, rowSums2, colMeans2, rowMeans2 = colSums, rowSums, colMeans, rowMeans

    def colVars(self, x, na_rm=False, center=None, dims=1):
        return self._colStats("var1", x, na_rm, center, dims)

    def colSds(self, x, na_rm=False, center=None, dims=1):
        return self._colStats("sd1", x, na_rm, center, dims)

    def rowVars(self, x, na_rm=False, center=None, dims=1):
        # :645-660
        if x.na_background:
            return self._rowStats("var1", x, na_rm, dims=dims)     # raises: no NaArray method
        nvals = self._rowCountVals(x, na_rm, dims)
        with np.errstate(all="ignore"):
            if center is None:
                center = self.rowSums(x, na_rm, dims) / nvals
            cx2 = self._rowStats("centered_X2_sum", x, na_rm, center, dims)
            return cx2 / (nvals - 1)

    def rowSds(self, x, na_rm=False, center=None, dims=1):
        with np.errstate(all="ignore"):
            return np.sqrt(self.rowVars(x, na_rm, center, dims))

    # ------------------------------------------------------------------
    # whole-array summarization  (R/SparseArray-summarization.R)
    # ------------------------------------------------------------------
    def summarize_SVT(self, op, x, na_rm=False, center=None):
        center = NA_real if center is None else float(center)
        ans, warn = self.SparseArray_Call("C_summarize_SVT", x, op,
                                          bool(na_rm), center)
        if warn:
            warnings.warn("NAs introduced by coercion of "
                          "infinite values to integers")
        return ans

    def anyNA(self, x): return self.summarize_SVT("anyNA", x)
    def countNAs(self, x): return self.summarize_SVT("countNAs", x)
    def any(self, x, na_rm=False): return self.summarize_SVT("any", x, na_rm)
    def all(self, x, na_rm=False): return self.summarize_SVT("all", x, na_rm)
    def min(self, x, na_rm=False): return self.summarize_SVT("min", x, na_rm)
    def max(self, x, na_rm=False): return self.summarize_SVT("max", x, na_rm)
    def range(self, x, na_rm=False): return self.summarize_SVT("range", x, na_rm)
    def sum(self, x, na_rm=False): return self.summarize_SVT("sum", x, na_rm)
    def prod(self, x, na_rm=False): return self.summarize_SVT("prod", x, na_rm)
    def mean(self, x, na_rm=False): return self.summarize_SVT("mean", x, na_rm)
    def var(self, x, na_rm=False): return self.summarize_SVT("var1", x, na_rm)
    def sd(self, x, na_rm=False): return self.summarize_SVT("sd1", x, na_rm)

    # ------------------------------------------------------------------
    # rowsum / colsum  (R/rowsum-methods.R)
    # ------------------------------------------------------------------
    @staticmethod
    def _compute_ugroup(group, expected_len, reorder):
        group = list(group)
        if len(group) != expected_len:
            raise SparseArrayError("incorrect length for 'group'")
        ug = list(dict.fromkeys(group))
        if reorder:
            ug = sorted(ug, key=lambda g: (g is None, g))
        return ug

    @staticmethod
    def _match(group, ugroup):
        pos = {g: i + 1 for i, g in enumerate(ugroup)}
        return np.asarray([pos[g] for g in group], dtype=np.int32)

    def rowsum(self, x, group, reorder=True, na_rm=False):
        """Returns (matrix ngroup x ncol, ugroup)."""
        self._no_NaArray("rowsum", x)
        if isinstance(x, SVT_SparseArray):
            nrow = x.dim[0]
        else:
            nrow = x[0][0]
        ugroup = self._compute_ugroup(group, nrow, reorder)
        g = self._match(group, ugroup)
        if isinstance(x, SVT_SparseArray):
            ans, ovflow = self.SparseArray_Call("C_rowsum_SVT", x, g,
                                                len(ugroup), bool(na_rm))
            if ovflow:
                warnings.warn("NAs produced by integer overflow")
        else:
            ans = self.SparseArray_Call("C_rowsum_dgCMatrix", x, g,
                                        len(ugroup), bool(na_rm))
        return ans, ugroup

    def colsum(self, x, group, reorder=True, na_rm=False):
        """Returns (matrix nrow x ngroup, ugroup)."""
        self._no_NaArray("colsum", x)
        if isinstance(x, SVT_SparseArray):
            ncol = x.dim[1]
        else:
            ncol = x[0][1]
        ugroup = self._compute_ugroup(group, ncol, reorder)
        g = self._match(group, ugroup)
        if isinstance(x, SVT_SparseArray):
            ans, ovflow = self.SparseArray_Call("C_colsum_SVT", x, g,
                                                len(ugroup), bool(na_rm))
            if ovflow:
                warnings.warn("NAs produced by integer overflow")
        else:
            ans = self.SparseArray_Call("C_colsum_dgCMatrix", x, g,
                                        len(ugroup), bool(na_rm))
        return ans, ugroup


    # ------------------------------------------------------------------
    # column statistics of dgCMatrix objects  (R/sparseMatrix-utils.R:300-330)
    # ``x`` = ((nrow, ncol), p, i, x) -- the dgCMatrix slots; colnames are not propagated
    # ------------------------------------------------------------------
    def _dgc_colstat(self, name, x, na_rm):
        if not (isinstance(x, tuple) and len(x) == 4):
            raise SparseArrayError("is(x, \"dgCMatrix\") is not TRUE")
        if not isinstance(na_rm, (bool, np.bool_)):
            raise SparseArrayError("'na.rm' must be TRUE or FALSE")
        return self.SparseArray_Call(name, x, bool(na_rm))

    def colMins_dgCMatrix(self, x, na_rm=False):
        return self._dgc_colstat("C_colMins_dgCMatrix", x, na_rm)

    def colMaxs_dgCMatrix(self, x, na_rm=False):
        return self._dgc_colstat("C_colMaxs_dgCMatrix", x, na_rm)

    def colRanges_dgCMatrix(self, x, na_rm=False):
        return self._dgc_colstat("C_colRanges_dgCMatrix", x, na_rm)

    def colVars_dgCMatrix(self, x, na_rm=False):
        return self._dgc_colstat("C_colVars_dgCMatrix", x, na_rm)


# ---------------------------------------------------------------------------
# Shared helpers for dispatchers (argument packing for the C ABIs)
# ---------------------------------------------------------------------------
OPCODES = {
    "anyNA": 1, "countNAs": 2, "any": 3, "all": 4, "min": 5, "max": 6,
    "range": 7, "sum": 8, "prod": 9, "mean": 10, "centered_X2_sum": 11,
    "sum_X_X2": 12, "var1": 13, "var2": 14, "sd1": 15, "sd2": 16,
}


def back_to_int(x: float) -> int:
    # BACK_TO_INT, src/Rvector_summarization.c:1199
    return int(x + 0.5) if x >= 0 else int(x - 0.5)


def naked_result(op: str, in_type: str, out_d, out_i):
    """res2nakedSEXP(), src/Rvector_summarization.c:1239-1296."""
    INT_MAX = 2 ** 31 - 1
    if op in ("anyNA", "any", "all"):
        return np.int32(out_i[0])
    if op == "countNAs":
        return np.float64(out_d[0]) if out_d[0] > INT_MAX else np.int32(back_to_int(out_d[0]))
    if op in ("min", "max") and in_type != "double":
        return np.int32(out_i[0])
    if op == "range":
        if in_type == "double":
            return np.array([out_d[0], out_d[1]], dtype=np.float64)
        return np.array([out_i[0], out_i[1]], dtype=np.int32)
    if op in ("sum", "prod") and in_type in ("logical", "integer"):
        v = out_d[0]
        if np.isnan(v):
            return NA_integer
        if v < -INT_MAX or v > INT_MAX:
            return np.float64(v)
        return np.int32(back_to_int(v))
    return np.float64(out_d[0])
