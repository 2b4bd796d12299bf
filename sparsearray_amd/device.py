"""Device-resident operands (HBM) and the device-level entry points.

torch is used here as plumbing only: it owns the device allocations and the
HIP stream the kernels are launched on.  Every computation is a call into
libsvt_hip.so (include/svt_hip.h, "device level").
"""
from __future__ import annotations

import ctypes
from ctypes import c_double, c_int, c_int64, c_size_t, c_void_p

import numpy as np
import torch

from . import _hip
from .api import OPCODES, SparseArrayError, SparseArrayUnsupported
from .svt import INTSXP, LGLSXP, REALSXP

_protos_done = False


def _lib():
    global _protos_done
    lib = _hip.init()
    if not _protos_done:
        lib.svt_wrap_device_csc.restype = c_void_p
        lib.svt_wrap_device_csc.argtypes = [c_int, c_int64, c_int64, c_int64,
                                            c_void_p, c_void_p, c_void_p]
        lib.svt_release.argtypes = [c_void_p]
        lib.svt_release.restype = None
        lib.svt_dev_crossprod_ws_bytes.restype = c_size_t
        lib.svt_dev_crossprod_ws_bytes.argtypes = [c_int64, c_int64, c_int]
        lib.svt_dev_dense_prepare.argtypes = [c_void_p, c_int64, c_int64, c_int, c_int,
                                              c_int, c_void_p, c_size_t, c_void_p]
        lib.svt_dev_crossprod_prepared.argtypes = [c_void_p, c_void_p, c_int, c_void_p,
                                                   c_int64, c_int64, c_void_p]
        lib.svt_dev_crossprod_csc_dense.argtypes = [c_void_p, c_void_p, c_int64, c_int,
                                                    c_int, c_void_p, c_int64, c_int64,
                                                    c_void_p, c_size_t, c_void_p]
        lib.svt_dev_pbc_build.restype = c_void_p
        lib.svt_dev_pbc_build.argtypes = [c_void_p, c_int, c_int, c_int]
        lib.svt_dev_pbc_release.argtypes = [c_void_p]
        lib.svt_dev_pbc_release.restype = None
        lib.svt_dev_pbc_set_spare_cus.argtypes = [c_int]
        lib.svt_dev_pbc_set_spare_cus.restype = None
        lib.svt_dev_pbc_spare_cus.restype = c_int
        lib.svt_dev_pbc_set_gather_pacing.argtypes = [c_int, c_int]
        lib.svt_dev_pbc_set_gather_pacing.restype = None
        lib.svt_dev_crossprod_pbc_ws_bytes.restype = c_size_t
        lib.svt_dev_crossprod_pbc_ws_bytes.argtypes = [c_void_p, c_int]
        lib.svt_dev_crossprod_pbc.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                              c_int, c_void_p, c_int64, c_int64, c_void_p,
                                              c_size_t, c_void_p]
        lib.svt_dev_crossprod_pbc_phase.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                                    c_int, c_void_p, c_int64, c_int64, c_void_p,
                                                    c_size_t, c_void_p, c_int]
        lib.svt_dev_colstats.argtypes = [c_void_p, c_int, c_int, c_double, c_int64,
                                         c_void_p, c_void_p, c_void_p]
        lib.svt_dev_matmul_csc_csc_ws_bytes.restype = c_size_t
        lib.svt_dev_matmul_csc_csc_ws_bytes.argtypes = [c_void_p]
        lib.svt_dev_matmul_csc_csc.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_size_t,
                                               c_void_p, c_void_p]
        lib.svt_dev_matmul_csc_csc_prepare.argtypes = [c_void_p, c_void_p, c_size_t, c_void_p]
        lib.svt_dev_matmul_csc_csc_prepared.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_size_t,
                                                        c_void_p, c_void_p]
        lib.svt_dev_crossprod_csc_csc_ws_bytes.restype = c_size_t
        lib.svt_dev_crossprod_csc_csc_ws_bytes.argtypes = [c_void_p]
        lib.svt_dev_crossprod_csc_csc.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_size_t,
                                                  c_void_p, c_void_p]
        lib.svt_dev_crossprod_csc_csc_set_panel.argtypes = [c_int, c_int]
        lib.svt_dev_crossprod_csc_csc_set_panel.restype = None
        lib.svt_dev_colmedians_ws_bytes.restype = c_size_t
        lib.svt_dev_colmedians_ws_bytes.argtypes = [c_int64, c_int64]
        lib.svt_dev_colmedians.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]
        lib.svt_dev_rowstats_ws_bytes.restype = c_size_t
        lib.svt_dev_rowstats_ws_bytes.argtypes = [c_int64, c_int64]
        lib.svt_dev_rowsums.argtypes = [c_void_p, c_int, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]
        lib.svt_dev_rowsums_prepare.argtypes = [c_void_p, c_int64, c_void_p, c_size_t, c_void_p]
        lib.svt_dev_rowsums_prepared.argtypes = [c_void_p, c_int, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]
        lib.svt_dev_rowsum.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]
        lib.svt_dev_rowsum_gid_bytes.restype = c_size_t
        lib.svt_dev_rowsum_gid_bytes.argtypes = [c_void_p]
        lib.svt_dev_rowsum_prepare.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]
        lib.svt_dev_rowsum_prepared.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]
        lib.svt_colStats_out_Rtype.argtypes = [c_int, c_int]
        lib.svt_dev_transpose_ws_bytes.restype = c_size_t
        lib.svt_dev_transpose_ws_bytes.argtypes = [c_int64, c_int64]
        lib.svt_dev_transpose.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
        lib.svt_dev_aperm_ws_bytes.restype = c_size_t
        lib.svt_dev_aperm_ws_bytes.argtypes = [c_int64, c_int, c_void_p]
        lib.svt_dev_aperm.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_size_t, c_void_p]
        _protos_done = True
    return lib


def _check(rc):
    if rc != 0:
        raise (SparseArrayUnsupported if rc > 0 else SparseArrayError)(_lib().svt_last_error().decode())


def _stream() -> c_void_p:
    return c_void_p(torch.cuda.current_stream().cuda_stream)


class DeviceCSC:
    """An SVT in its device layout: col_ptr int64[ncol+1], row_idx int32[nnz],
    val f64|i32[nnz] (struct svt_dev_csc).  ``ncol`` counts leaves."""

    def __init__(self, nrow: int, col_ptr: torch.Tensor, row_idx: torch.Tensor,
                 val: torch.Tensor, logical: bool = False):
        assert col_ptr.dtype == torch.int64 and row_idx.dtype == torch.int32
        assert val.dtype in (torch.float64, torch.int32)
        assert col_ptr.is_cuda and row_idx.is_cuda and val.is_cuda
        self.nrow = int(nrow)
        self.ncol = int(col_ptr.numel() - 1)
        self.nnz = int(row_idx.numel())
        self.col_ptr, self.row_idx, self.val = col_ptr.contiguous(), row_idx.contiguous(), val.contiguous()
        self.Rtype = REALSXP if val.dtype == torch.float64 else (LGLSXP if logical else INTSXP)
        self._h = _lib().svt_wrap_device_csc(self.Rtype, self.nrow, self.ncol, self.nnz,
                                             self.col_ptr.data_ptr(), self.row_idx.data_ptr(),
                                             self.val.data_ptr())

    @classmethod
    def from_host(cls, nrow, col_ptr, row_idx, val, device="cuda"):
        return cls(nrow, torch.as_tensor(np.asarray(col_ptr, np.int64), device=device),
                   torch.as_tensor(np.asarray(row_idx, np.int32), device=device),
                   torch.as_tensor(np.asarray(val), device=device))

    @property
    def handle(self):
        return c_void_p(self._h)

    def t(self) -> "DeviceCSC":
        """t(x) on the device (2-d operands)."""
        dev = self.val.device
        cp = torch.empty(self.nrow + 1, dtype=torch.int64, device=dev)
        ri = torch.empty(self.nnz, dtype=torch.int32, device=dev)
        vv = torch.empty(self.nnz, dtype=self.val.dtype, device=dev)
        ws = torch.empty(_lib().svt_dev_transpose_ws_bytes(self.nrow, self.nnz), dtype=torch.uint8, device=dev)
        _check(_lib().svt_dev_transpose(self.handle, cp.data_ptr(), ri.data_ptr(), vv.data_ptr(),
                                        ws.data_ptr(), ws.numel(), _stream()))
        return DeviceCSC(self.ncol, cp, ri, vv, logical=self.Rtype == LGLSXP)

    def aperm(self, dim, perm):
        """aperm(x, perm) on the device for the N-d array of extents ``dim`` stored in
        this layout (dim[0] == nrow, prod(dim[1:]) == ncol); ``perm`` is 1-based.
        Returns (DeviceCSC of the permuted array, its dim)."""
        dim = np.asarray(dim, dtype=np.int64)
        perm = np.asarray(perm, dtype=np.int32)
        new_dim = tuple(int(dim[p - 1]) for p in perm)
        new_nl = int(np.prod(new_dim[1:], dtype=np.int64)) if len(new_dim) > 1 else 1
        dev = self.val.device
        cp = torch.empty(new_nl + 1, dtype=torch.int64, device=dev)
        ri = torch.empty(self.nnz, dtype=torch.int32, device=dev)
        vv = torch.empty(self.nnz, dtype=self.val.dtype, device=dev)
        nb = _lib().svt_dev_aperm_ws_bytes(self.nnz, len(dim), dim.ctypes.data)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        _check(_lib().svt_dev_aperm(self.handle, len(dim), dim.ctypes.data, perm.ctypes.data,
                                    cp.data_ptr(), ri.data_ptr(), vv.data_ptr(), ws.data_ptr(),
                                    ws.numel(), _stream()))
        return DeviceCSC(new_dim[0], cp, ri, vv, logical=self.Rtype == LGLSXP), new_dim

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib().svt_release(self._h)
                self._h = None
        except Exception:
            pass


class CrossprodPlan:
    """Reusable workspace for crossprod(A, Y) with K dense columns."""

    def __init__(self, A: DeviceCSC, K: int):
        self.A, self.K = A, int(K)
        n = _lib().svt_dev_crossprod_ws_bytes(A.nrow, A.ncol, self.K)
        self.ws = torch.empty(n, dtype=torch.uint8, device=A.val.device)

    def prepare(self, Y: torch.Tensor, ldY: int, tr_y: bool = False):
        """Y: device buffer holding the dense operand in R (column-major) layout."""
        _check(_lib().svt_dev_dense_prepare(Y.data_ptr(), ldY, self.A.nrow, self.K,
                                            int(tr_y), self.A.Rtype, self.ws.data_ptr(),
                                            self.ws.numel(), _stream()))

    def multiply(self, out: torch.Tensor, stride_c: int, stride_k: int):
        _check(_lib().svt_dev_crossprod_prepared(self.A.handle, self.ws.data_ptr(), self.K,
                                                 out.data_ptr(), stride_c, stride_k, _stream()))

    def run(self, Y, ldY, out, stride_c=1, stride_k=None, tr_y=False):
        if stride_k is None:
            stride_k = self.A.ncol
        self.prepare(Y, ldY, tr_y)
        self.multiply(out, stride_c, stride_k)


class PbcPlan:
    """Fast path of crossprod(A, Y) for f64: panel-blocked copy of A (built
    once, here) + workspace for K dense columns."""

    def __init__(self, A: DeviceCSC, K: int, CBW: int = 40, WPB: int = 16, logR: int = 7):
        # (0, 0, 0): the library picks the layout by density (LDS-DMA kernel or gather kernel)
        assert A.Rtype == REALSXP
        self.A, self.K = A, int(K)
        self._p = _lib().svt_dev_pbc_build(A.handle, CBW, WPB, logR)
        if not self._p:
            raise SparseArrayError(_lib().svt_last_error().decode())
        n = _lib().svt_dev_crossprod_pbc_ws_bytes(self._p, self.K)
        self.ws = torch.empty(n, dtype=torch.uint8, device=A.val.device)

    def run(self, Y, ldY, out, stride_c=1, stride_k=None, tr_y=False):
        if stride_k is None:
            stride_k = self.A.ncol
        _check(_lib().svt_dev_crossprod_pbc(self._p, self.A.handle, Y.data_ptr(), ldY, self.K,
                                            int(tr_y), out.data_ptr(), stride_c, stride_k,
                                            self.ws.data_ptr(), self.ws.numel(), _stream()))

    def run_phase(self, phase, Y, ldY, out, stride_c=1, stride_k=None, tr_y=False):
        if stride_k is None:
            stride_k = self.A.ncol
        _check(_lib().svt_dev_crossprod_pbc_phase(self._p, self.A.handle, Y.data_ptr(), ldY,
                                                  self.K, int(tr_y), out.data_ptr(), stride_c,
                                                  stride_k, self.ws.data_ptr(), self.ws.numel(),
                                                  _stream(), phase))

    def __del__(self):
        try:
            if getattr(self, "_p", None):
                _lib().svt_dev_pbc_release(self._p)
                self._p = None
        except Exception:
            pass


def aperm_route_counts(reset=False) -> dict:
    """Which route the transpositions / permutations of this process took so far (svt_dev_aperm_route_counts)."""
    names = ("t_bucketed", "t_key_sort", "leaf_preserving", "first_two_axes_swapped", "slab", "via_intermediate_3d",
             "general_composed", "key_sort_32", "key_sort_64", "slab_refused_at_run_time")
    buf = (c_int64 * 10)()
    _lib().svt_dev_aperm_route_counts.argtypes = [c_void_p, c_int]
    _lib().svt_dev_aperm_route_counts.restype = None
    _lib().svt_dev_aperm_route_counts(buf, int(bool(reset)))
    return dict(zip(names, (int(x) for x in buf)))


def trim_layout_pool() -> None:
    """Returns the memory the layout pools keep for the next build to the driver (svt_dev_pbc_trim)."""
    _lib().svt_dev_pbc_trim.restype = None
    _lib().svt_dev_pbc_trim()


def set_spare_cus(n: int) -> None:
    """CUs the LDS-DMA product kernel leaves idle from now on (0 = none; include/svt_hip.h:
    svt_dev_pbc_set_spare_cus) -- room for a collective's kernels beside the product."""
    _lib().svt_dev_pbc_set_spare_cus(int(n))


def spare_cus() -> int:
    return int(_lib().svt_dev_pbc_spare_cus())


def set_gather_pacing(dsync: int = 1, spin: int = 256) -> None:
    """Pacing of the gather product of very sparse operands (include/svt_hip.h:
    svt_dev_pbc_set_gather_pacing); dsync < 0 selects the unpaced kernels."""
    _lib().svt_dev_pbc_set_gather_pacing(int(dsync), int(spin))


def set_round_launches(on=True) -> None:
    """One launch per round of workgroups for products with many column blocks (svt_dev_pbc_set_round_launches):
    False / 0 = one launch, True / 1 = per round with the partly filled last round cut by rows (default), 2 = per
    round with the last round whole."""
    _lib().svt_dev_pbc_set_round_launches.restype = None
    _lib().svt_dev_pbc_set_round_launches(int(on))


def crossprod_csc_dense(A: DeviceCSC, Y: torch.Tensor) -> torch.Tensor:
    """crossprod(A, Y) for a dense Y given as a (K, nrow) C-contiguous tensor,
    i.e. the column-major nrow x K matrix R would hand over.  Returns the
    (K, ncol) C-contiguous tensor that is the column-major ncol x K result."""
    K, nrow = Y.shape
    assert nrow == A.nrow and Y.is_contiguous()
    out = torch.zeros((K, A.ncol), dtype=torch.float64, device=Y.device)
    CrossprodPlan(A, K).run(Y, nrow, out)
    return out


def colstats(A: DeviceCSC, op: str, na_rm=False, center=float("nan"), inner=1):
    oc = OPCODES[op]
    rt = _lib().svt_colStats_out_Rtype(oc, A.Rtype)
    nseg = A.ncol // inner
    out = torch.empty(nseg, dtype=torch.float64 if rt == REALSXP else torch.int32,
                      device=A.val.device)
    warn = torch.zeros(4, dtype=torch.int32, device=A.val.device)
    _check(_lib().svt_dev_colstats(A.handle, oc, int(na_rm), float(center), inner,
                                   out.data_ptr(), warn.data_ptr(), _stream()))
    return out, warn


def colmedians(A: DeviceCSC, na_rm=False, out=None, ws=None):
    """colMedians() of a resident 2-D operand (include/svt_hip.h, svt_dev_colmedians)."""
    if out is None:
        out = torch.empty(A.ncol, dtype=torch.float64, device=A.val.device)
    if ws is None:
        ws = torch.empty(_lib().svt_dev_colmedians_ws_bytes(A.nnz, A.ncol), dtype=torch.uint8,
                         device=A.val.device)
    _check(_lib().svt_dev_colmedians(A.handle, int(na_rm), out.data_ptr(), ws.data_ptr(),
                                     ws.numel(), _stream()))
    return out


def matmul_csc_csc(A: DeviceCSC, B: DeviceCSC, out=None, ws=None):
    """A %*% B for two resident sparse operands, B much sparser than a dense matrix (include/svt_hip.h,
    svt_dev_matmul_csc_csc).  Returns (out, not_finite): out is the (B.ncol, A.nrow) C-contiguous tensor that is
    the column-major A.nrow x B.ncol matrix; not_finite is a device int32 tensor, nonzero when a non-finite value
    or an NA took part -- the result then has to come from the dense route."""
    assert A.ncol == B.nrow
    dev = A.val.device
    if out is None:
        out = torch.empty((B.ncol, A.nrow), dtype=torch.float64, device=dev)
    if ws is None:
        ws = torch.empty(_lib().svt_dev_matmul_csc_csc_ws_bytes(A.handle), dtype=torch.uint8, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    _check(_lib().svt_dev_matmul_csc_csc(A.handle, B.handle, out.data_ptr(), A.nrow, ws.data_ptr(), ws.numel(),
                                         flag.data_ptr(), _stream()))
    return out, flag


def crossprod_csc_csc(Xt: DeviceCSC, Y: DeviceCSC, sym=False, out=None, ws=None):
    """crossprod(X, Y) of two resident sparse operands without a dense buffer (include/svt_hip.h,
    svt_dev_crossprod_csc_csc): ``Xt`` is t(X) (``X.t()``), ``sym`` says Y is X.  Returns (out, not_finite): out is the
    (ncol(Y), ncol(X)) C-contiguous tensor that is the column-major ncol(X) x ncol(Y) matrix; not_finite a device
    int32 tensor, nonzero when a non-finite value or an NA took part -- the result then has to come from the
    dense-buffer route."""
    assert Xt.ncol == Y.nrow
    dev = Y.val.device
    if out is None:
        out = torch.empty((Y.ncol, Xt.nrow), dtype=torch.float64, device=dev)
    if ws is None:
        ws = torch.empty(_lib().svt_dev_crossprod_csc_csc_ws_bytes(Xt.handle), dtype=torch.uint8, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    _check(_lib().svt_dev_crossprod_csc_csc(Xt.handle, Y.handle, int(bool(sym)), out.data_ptr(), Xt.nrow,
                                            ws.data_ptr(), ws.numel(), flag.data_ptr(), _stream()))
    return out, flag


def crossprod_csc_csc_dense_buffer(X: DeviceCSC, Y: DeviceCSC, out=None):
    """The dense-buffer route of crossprod(X, Y) on resident operands (svt_dev_crossprod_csc_csc_dense_buffer;
    ``Y is X``: the unary form).  Allocates and synchronises inside.  Returns the (ncol(Y), ncol(X)) C-contiguous
    tensor that is the column-major result."""
    _lib().svt_dev_crossprod_csc_csc_dense_buffer.argtypes = [c_void_p, c_void_p, c_void_p]
    if out is None:
        out = torch.empty((Y.ncol, X.ncol), dtype=torch.float64, device=Y.val.device)
    torch.cuda.synchronize()
    _check(_lib().svt_dev_crossprod_csc_csc_dense_buffer(X.handle, X.handle if Y is X else Y.handle, out.data_ptr()))
    return out


def set_sparse_crossprod_cost(factor=1.0) -> None:
    """Route choice of the host entry points crossprod(x) / crossprod(x, y) (svt_sparse_crossprod_set_cost):
    < 0 never the sparse-aware kernel, 0 always, 1 the measured model."""
    _lib().svt_sparse_crossprod_set_cost.argtypes = [c_double]
    _lib().svt_sparse_crossprod_set_cost.restype = None
    _lib().svt_sparse_crossprod_set_cost(float(factor))


def set_sparse_crossprod_panel(one_block_max=-1, log2_panel=-1) -> None:
    """Cell-panel shape of crossprod_csc_csc() (svt_dev_crossprod_csc_csc_set_panel); defaults restored by -1."""
    _lib().svt_dev_crossprod_csc_csc_set_panel(int(one_block_max), int(log2_panel))


class SpmmPlan:
    """What `A %*% B` (both sparse) needs from A alone -- the table of run bounds per row panel and the scan of
    its values -- done once (svt_dev_matmul_csc_csc_prepare), as `PbcPlan` does for crossprod(A, Y)."""

    def __init__(self, A: DeviceCSC):
        self.A = A
        self.ws = torch.empty(_lib().svt_dev_matmul_csc_csc_ws_bytes(A.handle), dtype=torch.uint8, device=A.val.device)
        _check(_lib().svt_dev_matmul_csc_csc_prepare(A.handle, self.ws.data_ptr(), self.ws.numel(), _stream()))

    def run(self, B: DeviceCSC, out=None):
        """Returns (out, not_finite) like matmul_csc_csc(); the flag tensor is this call's own."""
        A = self.A
        assert A.ncol == B.nrow
        if out is None:
            out = torch.empty((B.ncol, A.nrow), dtype=torch.float64, device=A.val.device)
        # (products are asynchronous: a flag tensor shared between runs would show a later product's verdict to
        # whoever reads an earlier one late)
        flag = torch.zeros(1, dtype=torch.int32, device=A.val.device)
        _check(_lib().svt_dev_matmul_csc_csc_prepared(A.handle, B.handle, out.data_ptr(), A.nrow, self.ws.data_ptr(),
                                                      self.ws.numel(), flag.data_ptr(), _stream()))
        return out, flag


def rowsums(A: DeviceCSC, na_rm=False, inner=1, out=None, ws=None):
    if out is None:
        out = torch.empty(inner * A.nrow, dtype=torch.float64, device=A.val.device)
    if ws is None:
        ws = torch.empty(_lib().svt_dev_rowstats_ws_bytes(A.nrow, A.ncol), dtype=torch.uint8,
                         device=A.val.device)
    _check(_lib().svt_dev_rowsums(A.handle, int(na_rm), inner, out.data_ptr(), ws.data_ptr(),
                                  ws.numel(), _stream()))
    return out


class RowSumsPlan:
    """rowSums() of a resident operand with the table of run bounds built once (svt_dev_rowsums_prepare)."""

    def __init__(self, A: DeviceCSC, inner=1):
        self.A, self.inner = A, int(inner)
        self.ws = torch.empty(_lib().svt_dev_rowstats_ws_bytes(A.nrow, A.ncol), dtype=torch.uint8, device=A.val.device)
        _check(_lib().svt_dev_rowsums_prepare(A.handle, self.inner, self.ws.data_ptr(), self.ws.numel(), _stream()))

    def run(self, na_rm=False, out=None):
        A = self.A
        if out is None:
            out = torch.empty(self.inner * A.nrow, dtype=torch.float64, device=A.val.device)
        _check(_lib().svt_dev_rowsums_prepared(A.handle, int(na_rm), self.inner, out.data_ptr(), self.ws.data_ptr(),
                                               self.ws.numel(), _stream()))
        return out


def rowsum(A: DeviceCSC, group: torch.Tensor, ngroup: int, na_rm=False, out=None):
    assert group.dtype == torch.int32 and group.numel() == A.nrow
    if out is None:
        out = torch.empty((A.ncol, ngroup), dtype=torch.float64, device=A.val.device)
    _check(_lib().svt_dev_rowsum(A.handle, group.data_ptr(), int(ngroup), int(na_rm),
                                 out.data_ptr(), _stream()))
    return out


class RowsumPlan:
    """rowsum(A, group) for a pair used more than once: the 16-bit group id of every nonzero is computed once
    (svt_dev_rowsum_prepare), a call then streams 10 bytes per nonzero and looks nothing up
    (svt_dev_rowsum_prepared; src/rowsum_methods.c:44-64 for the rules)."""

    def __init__(self, A: DeviceCSC, group: torch.Tensor, ngroup: int):
        assert group.dtype == torch.int32 and group.numel() == A.nrow
        self.A, self.ngroup = A, int(ngroup)
        self.gid = torch.empty(_lib().svt_dev_rowsum_gid_bytes(A.handle), dtype=torch.uint8, device=A.val.device)
        _check(_lib().svt_dev_rowsum_prepare(A.handle, group.data_ptr(), self.ngroup, self.gid.data_ptr(),
                                             self.gid.numel(), _stream()))

    def run(self, na_rm=False, out=None):
        """(ncol, ngroup) C-contiguous = the column-major ngroup x ncol result, like rowsum()."""
        A = self.A
        if out is None:
            out = torch.empty((A.ncol, self.ngroup), dtype=torch.float64, device=A.val.device)
        _check(_lib().svt_dev_rowsum_prepared(A.handle, self.gid.data_ptr(), self.ngroup, int(na_rm),
                                              out.data_ptr(), _stream()))
        return out
