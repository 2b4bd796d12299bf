"""ctypes packing of the ``.Call``-level entry points onto a C ABI.

``CAbiDispatcher(lib, prefix)`` turns ``dispatcher("C_colStats_SVT", ...)``
into a call of ``<prefix>colStats_SVT`` in ``lib``.  The product instantiates
it once, for the HIP library (``libsvt_hip.so``, prefix ``svt_``;
include/svt_hip.h).  The packer is parameterised by (library, prefix) so that
the test-suite can drive its CPU checker -- which exposes the same host-level
signatures under another prefix -- with byte-identical arguments; nothing in
this package loads or names that checker.
"""
from __future__ import annotations

import ctypes
from ctypes import POINTER, byref, c_char_p, c_double, c_int, c_void_p

import numpy as np

from .api import OPCODES, SparseArrayError, SparseArrayUnsupported, naked_result
from .svt import (INTSXP, LGLSXP, REALSXP, SVT_SparseArray, make_view,
                  r_type_of, svt_view)

_RT = {"logical": LGLSXP, "integer": INTSXP, "double": REALSXP}


def _F(a: np.ndarray) -> np.ndarray:
    return np.asfortranarray(a)


def _ptr(a: np.ndarray):
    return ctypes.c_void_p(a.ctypes.data)


class CAbiDispatcher:
    def __init__(self, lib: ctypes.CDLL, prefix: str):
        self.lib = lib
        self.prefix = prefix
        self._declare()

    # -- prototypes ----------------------------------------------------------
    def _fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def _declare(self):
        V = POINTER(svt_view)
        I = c_int
        P = c_void_p
        protos = {
            "last_error": (c_char_p, []),
            "crossprod2_SVT_mat": (I, [V, P, I, I, I, I, P]),
            "crossprod2_mat_SVT": (I, [P, I, I, I, V, I, P]),
            "crossprod2_SVT_SVT": (I, [V, V, P]),
            "crossprod1_SVT": (I, [V, P]),
            "summarize_SVT": (I, [V, I, I, c_double, P, P, POINTER(I), POINTER(I)]),
            "colStats_out_Rtype": (I, [I, I]),
            "colStats_SVT": (I, [V, I, I, c_double, I, P, POINTER(I)]),
            "rowStats_SVT": (I, [V, I, I, P, I, P, POINTER(I)]),
            "rowsum_SVT": (I, [V, P, I, I, P, POINTER(I)]),
            "colsum_SVT": (I, [V, P, I, I, P, POINTER(I)]),
            "rowsum_dgCMatrix": (I, [I, I, P, P, P, P, I, I, P]),
            "colsum_dgCMatrix": (I, [I, I, P, P, P, P, I, I, P]),
        }
        for name in ("colMins_dgCMatrix", "colMaxs_dgCMatrix", "colRanges_dgCMatrix",
                     "colVars_dgCMatrix"):
            protos[name] = (I, [I, I, P, P, I, P])
        if hasattr(self.lib, self.prefix + "colMedians_SVT"):      # HIP library (the oracle's is Python)
            protos["colMedians_SVT"] = (I, [V, I, P])
            protos["rowMedians_SVT"] = (I, [V, I, P])
        # x %*% y in one call (device-side transposition): HIP library only
        for name, sig in (("matmul_SVT_mat", (I, [V, P, I, I, I, P])),
                          ("matmul_SVT_SVT", (I, [V, V, P])),
                          ("tcrossprod1_SVT", (I, [V, P])),
                          ("tcrossprod2_SVT_SVT", (I, [V, V, P]))):
            if hasattr(self.lib, self.prefix + name):
                protos[name] = sig
        for name, (res, args) in protos.items():
            f = self._fn(name)
            f.restype = res
            f.argtypes = args

    def _check(self, rc):
        if rc != 0:
            msg = self._fn("last_error")()
            raise (SparseArrayUnsupported if rc > 0 else SparseArrayError)(msg.decode() if msg else f"error {rc}")

    # -- dispatcher ------------------------------------------------------------
    def __call__(self, name: str, *args):
        if not name.startswith("C_"):
            raise SparseArrayError(f"unknown entry point {name}")
        return getattr(self, name)(*args)

    # thread control (src/thread_control.c:47-66) -------------------------------
    def C_get_num_procs(self):
        return int(self._fn("get_num_procs")())

    def C_get_max_threads(self):
        return int(self._fn("get_max_threads")())

    def C_set_max_threads(self, nthread: int):
        """Returns the previous value (the reference returns it too, R/thread-control.R:60-66)."""
        f = self._fn("set_max_threads")
        f.argtypes = [ctypes.c_int]
        return int(f(int(nthread)))

    # crossprod ---------------------------------------------------------------
    def C_crossprod2_SVT_mat(self, x: SVT_SparseArray, y: np.ndarray, tr_y: bool):
        y = _F(y)
        ans_ncol = y.shape[0] if tr_y else y.shape[1]
        out = np.zeros((x.dim[1], ans_ncol), dtype=np.float64, order="F")
        xv = make_view(x)
        self._check(self._fn("crossprod2_SVT_mat")(
            byref(xv), _ptr(y), y.shape[0], y.shape[1], _RT[r_type_of(y)],
            int(tr_y), _ptr(out)))
        return out

    def C_crossprod2_mat_SVT(self, x: np.ndarray, y: SVT_SparseArray, tr_x: bool):
        x = _F(x)
        ans_nrow = x.shape[0] if tr_x else x.shape[1]
        out = np.zeros((ans_nrow, y.dim[1]), dtype=np.float64, order="F")
        yv = make_view(y)
        self._check(self._fn("crossprod2_mat_SVT")(
            _ptr(x), x.shape[0], x.shape[1], _RT[r_type_of(x)], byref(yv),
            int(tr_x), _ptr(out)))
        return out

    def C_crossprod2_SVT_SVT(self, x: SVT_SparseArray, y: SVT_SparseArray):
        out = np.zeros((x.dim[1], y.dim[1]), dtype=np.float64, order="F")
        xv, yv = make_view(x), make_view(y)
        self._check(self._fn("crossprod2_SVT_SVT")(byref(xv), byref(yv), _ptr(out)))
        return out

    # colMedians (R/SparseArray-matrixStats.R:761-784) --------------------------------
    def C_colMedians_SVT(self, x: SVT_SparseArray, na_rm: bool):
        out = np.zeros(x.dim[1] if x.ndim == 2 else 0, dtype=np.float64)
        xv = make_view(x)
        self._check(self._fn("colMedians_SVT")(byref(xv), int(bool(na_rm)), _ptr(out)))
        return out

    def C_rowMedians_SVT(self, x: SVT_SparseArray, na_rm: bool):
        out = np.zeros(x.dim[0] if x.ndim == 2 else 0, dtype=np.float64)
        xv = make_view(x)
        self._check(self._fn("rowMedians_SVT")(byref(xv), int(bool(na_rm)), _ptr(out)))
        return out

    # resident operands (include/svt_hip.h; HIP library only) --------------------
    def resident_set_limit(self, nbytes: int):
        f = self._fn("resident_set_limit")
        f.argtypes = [ctypes.c_size_t]
        f.restype = c_int
        self._check(f(int(nbytes)))

    def resident_clear(self):
        f = self._fn("resident_clear")
        f.restype = None
        f()

    def resident_stats(self) -> dict:
        f = self._fn("resident_stats")
        f.restype = None
        b, e, h, m = ctypes.c_size_t(0), ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        f(byref(b), byref(e), byref(h), byref(m))
        return {"bytes": b.value, "entries": e.value, "hits": h.value, "misses": m.value}

    @property
    def accepts_mixed_types(self) -> bool:
        """The HIP library widens an integer operand of a mixed integer/double product on the
        device (include/svt_hip.h); the oracle keeps the reference's "same type" rule."""
        return self.prefix == "svt_"

    def has_entry(self, name: str) -> bool:
        return hasattr(self.lib, self.prefix + name[2:])

    def C_matmul_SVT_mat(self, x: SVT_SparseArray, y: np.ndarray):
        y = _F(y)
        out = np.zeros((x.dim[0], y.shape[1]), dtype=np.float64, order="F")
        xv = make_view(x)
        self._check(self._fn("matmul_SVT_mat")(
            byref(xv), _ptr(y), y.shape[0], y.shape[1], _RT[r_type_of(y)], _ptr(out)))
        return out

    def C_matmul_SVT_SVT(self, x: SVT_SparseArray, y: SVT_SparseArray):
        out = np.zeros((x.dim[0], y.dim[1]), dtype=np.float64, order="F")
        xv, yv = make_view(x), make_view(y)
        self._check(self._fn("matmul_SVT_SVT")(byref(xv), byref(yv), _ptr(out)))
        return out

    def C_tcrossprod1_SVT(self, x: SVT_SparseArray):
        out = np.zeros((x.dim[0], x.dim[0]), dtype=np.float64, order="F")
        xv = make_view(x)
        self._check(self._fn("tcrossprod1_SVT")(byref(xv), _ptr(out)))
        return out

    def C_tcrossprod2_SVT_SVT(self, x: SVT_SparseArray, y: SVT_SparseArray):
        out = np.zeros((x.dim[0], y.dim[0]), dtype=np.float64, order="F")
        xv, yv = make_view(x), make_view(y)
        self._check(self._fn("tcrossprod2_SVT_SVT")(byref(xv), byref(yv), _ptr(out)))
        return out

    def C_crossprod1_SVT(self, x: SVT_SparseArray):
        out = np.zeros((x.dim[1], x.dim[1]), dtype=np.float64, order="F")
        xv = make_view(x)
        self._check(self._fn("crossprod1_SVT")(byref(xv), _ptr(out)))
        return out

    # stats -------------------------------------------------------------------
    def _opcode(self, op: str) -> int:
        if op not in OPCODES:
            raise SparseArrayError(
                "'op' must be one of: " + ", ".join(f'"{k}"' for k in OPCODES))
        return OPCODES[op]

    def C_summarize_SVT(self, x, op, na_rm, center):
        oc = self._opcode(op)
        out_d = np.zeros(2, np.float64)
        out_i = np.zeros(2, np.int32)
        out_Rtype, warn = c_int(0), c_int(0)
        xv = make_view(x)
        self._check(self._fn("summarize_SVT")(
            byref(xv), oc, int(na_rm), float(center), _ptr(out_d), _ptr(out_i),
            byref(out_Rtype), byref(warn)))
        return naked_result(op, x.type, out_d, out_i), bool(warn.value)

    def _stat_out(self, op, x, shape):
        oc = self._opcode(op)
        rt = self._fn("colStats_out_Rtype")(oc, x.Rtype)
        if rt < 0:
            self._check(rt)
        dtype = np.float64 if rt == REALSXP else np.int32
        n = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
        flat = np.zeros(max(n, 1), dtype=dtype)
        return oc, flat, n

    def C_colStats_SVT(self, x, op, na_rm, center, dims):
        shape = tuple(x.dim[dims:])
        oc, flat, n = self._stat_out(op, x, shape)
        warn = c_int(0)
        xv = make_view(x)
        self._check(self._fn("colStats_SVT")(
            byref(xv), oc, int(na_rm), float(center), int(dims), _ptr(flat),
            byref(warn)))
        flat = flat[:n]
        ans = flat.reshape(shape, order="F") if len(shape) > 1 else flat
        return ans, bool(warn.value)

    def C_rowStats_SVT(self, x, op, na_rm, center, dims):
        shape = tuple(x.dim[:dims])
        oc, flat, n = self._stat_out(op, x, shape)
        warn = c_int(0)
        cptr = None
        if center is not None:
            center = np.ascontiguousarray(
                np.reshape(np.asarray(center, np.float64), -1, order="F"))
            if center.size != n:
                raise SparseArrayError("unexpected 'center' length")
            cptr = _ptr(center)
        xv = make_view(x)
        self._check(self._fn("rowStats_SVT")(
            byref(xv), oc, int(na_rm), cptr, int(dims), _ptr(flat), byref(warn)))
        flat = flat[:n]
        ans = flat.reshape(shape, order="F") if len(shape) > 1 else flat
        return ans, bool(warn.value)

    # rowsum / colsum -----------------------------------------------------------
    def _xsum(self, fname, x, group, ngroup, na_rm, shape):
        dtype = np.float64 if x.type == "double" else np.int32
        out = np.zeros(shape, dtype=dtype, order="F")
        group = np.ascontiguousarray(group, dtype=np.int32)
        ov = c_int(0)
        xv = make_view(x)
        self._check(self._fn(fname)(byref(xv), _ptr(group), int(ngroup),
                                    int(na_rm), _ptr(out), byref(ov)))
        return out, bool(ov.value)

    def C_rowsum_SVT(self, x, group, ngroup, na_rm):
        if x.ndim != 2:
            raise SparseArrayError("input object must have 2 dimensions")
        return self._xsum("rowsum_SVT", x, group, ngroup, na_rm,
                          (ngroup, x.dim[1]))

    def C_colsum_SVT(self, x, group, ngroup, na_rm):
        if x.ndim != 2:
            raise SparseArrayError("input object must have 2 dimensions")
        return self._xsum("colsum_SVT", x, group, ngroup, na_rm,
                          (x.dim[0], ngroup))

    def _dgc(self, fname, x, group, ngroup, na_rm, shape):
        (nrow, ncol), p, i, xx = x
        p = np.ascontiguousarray(p, np.int32)
        i = np.ascontiguousarray(i, np.int32)
        xx = np.ascontiguousarray(xx, np.float64)
        group = np.ascontiguousarray(group, dtype=np.int32)
        out = np.zeros(shape, dtype=np.float64, order="F")
        self._check(self._fn(fname)(int(nrow), int(ncol), _ptr(xx), _ptr(i),
                                    _ptr(p), _ptr(group), int(ngroup),
                                    int(na_rm), _ptr(out)))
        return out

    # aperm (src/SparseArray_aperm.c:935-970) ------------------------------------
    def C_aperm_SVT(self, x: SVT_SparseArray, perm):
        """Returns the permuted array as an SVT_SparseArray."""
        perm = np.asarray(perm, dtype=np.int32)
        if perm.size != x.ndim:
            raise SparseArrayError("'perm' must have one element per dimension")
        if sorted(perm.tolist()) != list(range(1, x.ndim + 1)):
            raise SparseArrayError(f"'perm' must be a permutation of 1:{x.ndim}")
        new_dim = tuple(x.dim[p - 1] for p in perm)
        nnz = x.nzcount()
        new_nl = int(np.prod(new_dim[1:], dtype=np.int64)) if x.ndim > 1 else 1
        cp = np.zeros(new_nl + 1, dtype=np.int64)
        ri = np.zeros(max(nnz, 1), dtype=np.int32)
        vv = np.zeros(max(nnz, 1), dtype=x.np_dtype)
        view = make_view(x)
        f = self._fn("aperm_SVT")
        f.restype = ctypes.c_int
        f.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        self._check(f(ctypes.addressof(view), perm.ctypes.data, cp.ctypes.data, ri.ctypes.data,
                      vv.ctypes.data))
        dn = None
        if x.dimnames is not None:
            dn = [x.dimnames[p - 1] for p in perm]
        ans = SVT_SparseArray.from_csc(new_dim, x.type, cp, ri[:nnz], vv[:nnz], dimnames=dn)
        ans.na_background = x.na_background
        return ans

    # column statistics of a dgCMatrix (src/sparseMatrix_utils.c:105-223) ---------------
    def _dgc_colstat(self, fname, x, na_rm, ncols_out):
        (nrow, ncol), p, _i, xx = x
        p = np.ascontiguousarray(p, np.int32)
        xx = np.ascontiguousarray(xx, np.float64)
        out = np.zeros((ncol, ncols_out) if ncols_out > 1 else ncol, dtype=np.float64, order="F")
        self._check(self._fn(fname)(int(nrow), int(ncol), _ptr(xx), _ptr(p), int(bool(na_rm)),
                                    _ptr(out)))
        return out

    def C_colMins_dgCMatrix(self, x, na_rm):
        return self._dgc_colstat("colMins_dgCMatrix", x, na_rm, 1)

    def C_colMaxs_dgCMatrix(self, x, na_rm):
        return self._dgc_colstat("colMaxs_dgCMatrix", x, na_rm, 1)

    def C_colRanges_dgCMatrix(self, x, na_rm):
        return self._dgc_colstat("colRanges_dgCMatrix", x, na_rm, 2)

    def C_colVars_dgCMatrix(self, x, na_rm):
        return self._dgc_colstat("colVars_dgCMatrix", x, na_rm, 1)

    # t() (src/SparseArray_aperm.c:395-423) ----------------------------------------------
    def C_transpose_2D_SVT(self, x: SVT_SparseArray):
        """Returns t(x) as an SVT_SparseArray (the glue rebuilds the R leaves the same way)."""
        if x.ndim != 2:
            raise SparseArrayError("object to transpose must have exactly 2 dimensions")
        nnz = x.nzcount()
        cp = np.zeros(x.dim[0] + 1, dtype=np.int64)
        ri = np.zeros(max(nnz, 1), dtype=np.int32)
        vv = np.zeros(max(nnz, 1), dtype=x.np_dtype)
        view = make_view(x)
        f = self._fn("transpose_2D_SVT")
        f.restype = ctypes.c_int
        f.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p]
        self._check(f(ctypes.addressof(view), cp.ctypes.data, ri.ctypes.data, vv.ctypes.data))
        dn = None
        if x.dimnames is not None:
            dn = [x.dimnames[1], x.dimnames[0]]
        ans = SVT_SparseArray.from_csc((x.dim[1], x.dim[0]), x.type, cp, ri[:nnz], vv[:nnz],
                                       dimnames=dn)
        ans.na_background = x.na_background
        return ans

    def C_rowsum_dgCMatrix(self, x, group, ngroup, na_rm):
        """``x`` = ((nrow, ncol), p, i, x) -- the dgCMatrix slots."""
        return self._dgc("rowsum_dgCMatrix", x, group, ngroup, na_rm,
                         (ngroup, x[0][1]))

    def C_colsum_dgCMatrix(self, x, group, ngroup, na_rm):
        return self._dgc("colsum_dgCMatrix", x, group, ngroup, na_rm,
                         (x[0][0], ngroup))
