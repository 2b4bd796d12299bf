"""Host-side mirror of the reference's SVT_SparseArray container.

Only what the compute hot path needs: the object model (dim / type / SVT
leaves, reference R/SVT_SparseArray-class.R:29-40, src/leaf_utils.h:10-31),
R's missing-value encodings, coercion from/to dense arrays (the way the
reference's tests build their inputs), 2-D transposition (``t()``), and the
flat "leaf table" that the C-ABI consumes.

Dense arrays use R index semantics: ``a[i, j, k]`` with dim 1 fastest when
flattened (Fortran order).
"""
from __future__ import annotations

import ctypes
import struct
from typing import List, Optional, Sequence, Tuple

import numpy as np

# ---------------------------------------------------------------------------
# R's missing values
# ---------------------------------------------------------------------------
NA_integer = np.int32(-2 ** 31)
NA_logical = NA_integer
NA_real = np.frombuffer(struct.pack("<Q", 0x7FF00000000007A2), dtype=np.float64)[0]

LGLSXP, INTSXP, REALSXP = 10, 13, 14
_RTYPE_OF = {"logical": LGLSXP, "integer": INTSXP, "double": REALSXP}
_TYPE_OF_R = {v: k for k, v in _RTYPE_OF.items()}
_NP_OF = {"logical": np.int32, "integer": np.int32, "double": np.float64}


def is_NA_real(x) -> np.ndarray:
    """R_IsNA(): NaN whose low word is 1954."""
    x = np.asarray(x, dtype=np.float64)
    lo = x.view(np.uint64) & np.uint64(0xFFFFFFFF)
    return np.isnan(x) & (lo == np.uint64(1954))


def is_NaN_real(x) -> np.ndarray:
    """R_IsNaN(): NaN that is not NA."""
    x = np.asarray(x, dtype=np.float64)
    return np.isnan(x) & ~is_NA_real(x)


def r_type_of(a: np.ndarray) -> str:
    if a.dtype == np.float64:
        return "double"
    if a.dtype == np.bool_:
        return "logical"
    if a.dtype == np.int32:
        return "integer"
    raise TypeError(f"unsupported dtype {a.dtype}; use float64, int32 or bool")


Leaf = Optional[Tuple[np.ndarray, Optional[np.ndarray]]]  # (nzoffs, nzvals|None)


class SVT_SparseArray:
    """An N-d sparse array stored as a flat list of leaves.

    ``leaves[j]`` is the sparse vector along dim 1 at outer position ``j``
    (column-major over dims 2..N): ``None`` for an empty leaf, else
    ``(nzoffs int32 ascending, nzvals)`` with ``nzvals is None`` for a lacunar
    leaf (all ones, src/leaf_utils.h:28-34).  ``svt_is_null`` mirrors
    ``x@SVT == NULL`` (all-zero array).  ``na_background=True`` makes the
    object an NaArray (R/NaArray-class.R): the implicit value is NA instead of
    zero and the leaves hold the non-NA entries (``R_IsNA`` for doubles: a NaN
    is stored, src/Rvector_utils.c:586-596); no lacunar leaves then.
    """

    def __init__(self, dim: Sequence[int], type: str, leaves: List[Leaf],
                 dimnames=None, svt_is_null: Optional[bool] = None,
                 na_background: bool = False):
        self.dim = tuple(int(d) for d in dim)
        if type not in _RTYPE_OF:
            raise ValueError(f"unsupported type {type!r}")
        self.type = type
        nleaves = int(np.prod(self.dim[1:], dtype=np.int64)) if len(self.dim) > 1 else 1
        if len(leaves) != nleaves:
            raise ValueError("wrong number of leaves")
        self.leaves = leaves
        self.dimnames = dimnames
        if svt_is_null is None:
            svt_is_null = all(lf is None for lf in leaves)
        self.svt_is_null = bool(svt_is_null)
        self.na_background = bool(na_background)
        self._keepalive = None

    # -- basic accessors -----------------------------------------------------
    @property
    def ndim(self) -> int:
        return len(self.dim)

    @property
    def Rtype(self) -> int:
        return _RTYPE_OF[self.type]

    @property
    def np_dtype(self):
        return _NP_OF[self.type]

    def nzcount(self) -> int:
        return sum(len(lf[0]) for lf in self.leaves if lf is not None)

    # -- coercion from / to dense -------------------------------------------
    @classmethod
    def from_dense(cls, a, type: Optional[str] = None, dimnames=None,
                   lacunar: bool = True, na_background: bool = False) -> "SVT_SparseArray":
        a = np.asarray(a)
        if type is None:
            type = r_type_of(a)
        if a.dtype == np.bool_:
            a = a.astype(np.int32)
        a = a.astype(_NP_OF[type], copy=False)
        if a.ndim == 0:
            raise ValueError("need at least 1 dimension")
        dim = a.shape
        nleaves = int(np.prod(dim[1:], dtype=np.int64))
        flat = np.reshape(a, (dim[0], nleaves), order="F")
        leaves: List[Leaf] = []
        for j in range(flat.shape[1]):
            col = flat[:, j]
            if na_background:
                keep = ~is_NA_real(col) if type == "double" else col != NA_integer
            else:
                keep = col != 0
            nz = np.flatnonzero(keep).astype(np.int32)
            if nz.size == 0:
                leaves.append(None)
                continue
            vals = np.ascontiguousarray(col[nz])
            if lacunar and not na_background and np.all(vals == 1):
                leaves.append((nz, None))
            else:
                leaves.append((nz, vals))
        return cls(dim, type, leaves, dimnames=dimnames, na_background=na_background)

    def to_dense(self) -> np.ndarray:
        n0 = self.dim[0]
        flat = np.zeros((n0, len(self.leaves)), dtype=self.np_dtype, order="F")
        if self.na_background:
            flat[...] = NA_real if self.type == "double" else NA_integer
        for j, lf in enumerate(self.leaves):
            if lf is None:
                continue
            offs, vals = lf
            flat[offs, j] = 1 if vals is None else vals
        return np.reshape(flat, self.dim, order="F")

    def with_type(self, type: str) -> "SVT_SparseArray":
        """``type(x) <- type`` for the promotions crossprod needs
        (R/SparseMatrix-mult.R:41-47): integer/logical -> double."""
        if type == self.type:
            return self
        if type != "double":
            raise ValueError("only promotion to \"double\" is supported")
        leaves: List[Leaf] = []
        for lf in self.leaves:
            if lf is None or lf[1] is None:
                leaves.append(lf)
                continue
            v = lf[1].astype(np.float64)
            v[lf[1] == NA_integer] = NA_real
            leaves.append((lf[0], v))
        return SVT_SparseArray(self.dim, "double", leaves, self.dimnames,
                               self.svt_is_null, self.na_background)

    def t(self) -> "SVT_SparseArray":
        """2-D transposition on the host, for building test and benchmark inputs only: the
        product's ``t()`` is ``Session.t`` = one ``C_transpose_2D_SVT`` call on the device
        (reference: src/SparseArray_aperm.c:348-423 -- count / allocate / scatter; a stable
        sort by row of the column-ordered nonzeros is that scatter)."""
        if self.ndim != 2:
            raise ValueError("t() needs a 2-D object")
        nrow, ncol = self.dim
        col_ptr, row_idx, val = self.to_csc()
        col_of = np.repeat(np.arange(ncol, dtype=np.int32), np.diff(col_ptr))
        order = np.argsort(row_idx, kind="stable")
        new_ptr = np.zeros(nrow + 1, dtype=np.int64)
        np.cumsum(np.bincount(row_idx, minlength=nrow), out=new_ptr[1:])
        new_idx, new_val = col_of[order], val[order]
        leaves: List[Leaf] = []
        for i in range(nrow):
            s, e = int(new_ptr[i]), int(new_ptr[i + 1])
            if s == e:
                leaves.append(None)
            elif not self.na_background and np.all(new_val[s:e] == 1):
                leaves.append((new_idx[s:e], None))
            else:
                leaves.append((new_idx[s:e], new_val[s:e]))
        dn = None
        if self.dimnames is not None:
            dn = [self.dimnames[1], self.dimnames[0]]
        return SVT_SparseArray((ncol, nrow), self.type, leaves, dn, na_background=self.na_background)

    # -- CSC marshalling (model: dump_SVT_to_CsparseMatrix_slots,
    #    src/SVT_SparseArray_class.c:598-633) ---------------------------------
    def to_csc(self):
        """(col_ptr int64[nleaves+1], row_idx int32[nnz], val[nnz]);
        lacunar leaves are expanded to explicit ones."""
        n = len(self.leaves)
        col_ptr = np.zeros(n + 1, dtype=np.int64)
        for j, lf in enumerate(self.leaves):
            col_ptr[j + 1] = col_ptr[j] + (0 if lf is None else len(lf[0]))
        nnz = int(col_ptr[-1])
        row_idx = np.empty(nnz, dtype=np.int32)
        val = np.empty(nnz, dtype=self.np_dtype)
        for j, lf in enumerate(self.leaves):
            if lf is None:
                continue
            s, e = col_ptr[j], col_ptr[j + 1]
            row_idx[s:e] = lf[0]
            val[s:e] = 1 if lf[1] is None else lf[1]
        return col_ptr, row_idx, val

    @classmethod
    def from_csc(cls, dim, type, col_ptr, row_idx, val, dimnames=None):
        """Leaves are views into the CSC arrays (no copies)."""
        col_ptr = np.asarray(col_ptr, dtype=np.int64)
        row_idx = np.ascontiguousarray(row_idx, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=_NP_OF[type])
        leaves: List[Leaf] = []
        for j in range(len(col_ptr) - 1):
            s, e = int(col_ptr[j]), int(col_ptr[j + 1])
            leaves.append(None if s == e else (row_idx[s:e], val[s:e]))
        return cls(dim, type, leaves, dimnames)


# ---------------------------------------------------------------------------
# The flat leaf table handed over the C-ABI (struct svt_view in
# include/svt_hip.h; the oracle uses the same layout).
# ---------------------------------------------------------------------------
class svt_view(ctypes.Structure):
    _fields_ = [
        ("Rtype", ctypes.c_int32),
        ("ndim", ctypes.c_int32),
        ("dim", ctypes.POINTER(ctypes.c_int32)),
        ("svt_is_null", ctypes.c_int32),
        ("nleaves", ctypes.c_int64),
        ("nzcount", ctypes.POINTER(ctypes.c_int32)),
        ("nzoffs", ctypes.POINTER(ctypes.c_void_p)),
        ("nzvals", ctypes.POINTER(ctypes.c_void_p)),
        ("na_background", ctypes.c_int32),
    ]


def make_view(x: SVT_SparseArray) -> svt_view:
    """Build the leaf table of ``x``.  The returned struct keeps the numpy
    buffers it points into alive through ``view._keep``."""
    n = len(x.leaves)
    dim = np.asarray(x.dim, dtype=np.int32)
    nzcount = np.zeros(max(n, 1), dtype=np.int32)
    offs_p = np.zeros(max(n, 1), dtype=np.uintp)
    vals_p = np.zeros(max(n, 1), dtype=np.uintp)
    keep = [dim, nzcount, offs_p, vals_p]
    for j, lf in enumerate(x.leaves):
        if lf is None:
            continue
        offs, vals = lf
        if offs.dtype != np.int32 or not offs.flags.c_contiguous:
            offs = np.ascontiguousarray(offs, dtype=np.int32)
            keep.append(offs)
        nzcount[j] = len(offs)
        offs_p[j] = offs.ctypes.data
        if vals is not None:
            if vals.dtype != x.np_dtype or not vals.flags.c_contiguous:
                vals = np.ascontiguousarray(vals, dtype=x.np_dtype)
                keep.append(vals)
            vals_p[j] = vals.ctypes.data
    v = svt_view()
    v.Rtype = x.Rtype
    v.ndim = x.ndim
    v.dim = dim.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    v.svt_is_null = int(x.svt_is_null)
    v.na_background = int(x.na_background)
    v.nleaves = n
    v.nzcount = nzcount.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    v.nzoffs = offs_p.ctypes.data_as(ctypes.POINTER(ctypes.c_void_p))
    v.nzvals = vals_p.ctypes.data_as(ctypes.POINTER(ctypes.c_void_p))
    v._keep = (keep, x)
    return v


def make_view_from_csc(dim, type, col_ptr, row_idx, val) -> svt_view:
    """Leaf table whose leaves alias a CSC triple (vectorised; used for the
    large inputs where per-leaf Python objects would be too slow)."""
    dim = np.asarray(dim, dtype=np.int32)
    col_ptr = np.ascontiguousarray(col_ptr, dtype=np.int64)
    row_idx = np.ascontiguousarray(row_idx, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=_NP_OF[type])
    n = len(col_ptr) - 1
    nzcount = np.diff(col_ptr).astype(np.int32)
    offs_p = (row_idx.ctypes.data + 4 * col_ptr[:-1]).astype(np.uintp)
    vals_p = (val.ctypes.data + val.itemsize * col_ptr[:-1]).astype(np.uintp)
    if n == 0:
        nzcount = np.zeros(1, np.int32)
        offs_p = np.zeros(1, np.uintp)
        vals_p = np.zeros(1, np.uintp)
    v = svt_view()
    v.Rtype = _RTYPE_OF[type]
    v.ndim = len(dim)
    v.dim = dim.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    v.svt_is_null = int(col_ptr[-1] == 0)
    v.na_background = 0
    v.nleaves = n
    v.nzcount = nzcount.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    v.nzoffs = offs_p.ctypes.data_as(ctypes.POINTER(ctypes.c_void_p))
    v.nzvals = vals_p.ctypes.data_as(ctypes.POINTER(ctypes.c_void_p))
    v._keep = (dim, col_ptr, row_idx, val, nzcount, offs_p, vals_p)
    return v
