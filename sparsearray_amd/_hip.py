"""ctypes binding of libsvt_hip.so (include/svt_hip.h).

There is deliberately no fallback here: a missing library, a missing symbol or
a box without an MI355X raises ``HipBackendError`` at first use.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SVT_HIP_TUNING=1: the tuning build (make -C sparsearray_amd/csrc TUNING=1) with the knobs of
# tools/tune_pbc.py compiled in; never what the product, the tests or the bench load.
LIB_PATH = os.path.join(_HERE, "libsvt_hip_tuning.so" if os.environ.get("SVT_HIP_TUNING") == "1"
                        else "libsvt_hip.so")

# Every symbol include/svt_hip.h declares (checked by tests/test_abi.py).
EXPORTS = [
    "svt_init", "svt_last_error", "svt_device_arch",
    "svt_crossprod2_SVT_mat", "svt_crossprod2_mat_SVT",
    "svt_crossprod2_SVT_SVT", "svt_crossprod1_SVT",
    "svt_matmul_SVT_mat", "svt_matmul_SVT_SVT", "svt_tcrossprod1_SVT", "svt_tcrossprod2_SVT_SVT",
    "svt_colMedians_SVT", "svt_rowMedians_SVT", "svt_dev_colmedians_ws_bytes", "svt_dev_colmedians",
    "svt_resident_set_limit", "svt_resident_clear", "svt_resident_stats", "svt_dev_pbc_bytes", "svt_dev_pbc_set_spare_cus", "svt_dev_pbc_spare_cus", "svt_dev_pbc_set_gather_pacing", "svt_dev_pbc_set_round_launches", "svt_dev_matmul_csc_csc_ws_bytes", "svt_dev_matmul_csc_csc", "svt_dev_rowsums_prepare", "svt_dev_rowsums_prepared", "svt_dev_rowsum_gid_bytes", "svt_dev_rowsum_prepare", "svt_dev_rowsum_prepared", "svt_dev_matmul_csc_csc_prepare", "svt_dev_matmul_csc_csc_prepared",
    "svt_dev_crossprod_csc_csc_ws_bytes", "svt_dev_crossprod_csc_csc", "svt_dev_crossprod_csc_csc_set_panel", "svt_sparse_crossprod_set_cost", "svt_dev_crossprod_csc_csc_dense_buffer",
    "svt_summarize_SVT", "svt_colStats_out_Rtype", "svt_colStats_SVT",
    "svt_rowStats_SVT", "svt_rowsum_SVT", "svt_colsum_SVT",
    "svt_rowsum_dgCMatrix", "svt_colsum_dgCMatrix",
    "svt_colMins_dgCMatrix", "svt_colMaxs_dgCMatrix", "svt_colRanges_dgCMatrix", "svt_colVars_dgCMatrix",
    "svt_upload", "svt_wrap_device_csc", "svt_release",
    "svt_dev_crossprod_ws_bytes", "svt_dev_crossprod_csc_dense",
    "svt_dev_dense_prepare", "svt_dev_crossprod_prepared",
    "svt_dev_pbc_build", "svt_dev_pbc_release", "svt_dev_pbc_trim",
    "svt_dev_crossprod_pbc_ws_bytes", "svt_dev_crossprod_pbc", "svt_dev_crossprod_pbc_phase", "svt_dev_crossprod_pbc_from",
    "svt_get_num_procs", "svt_get_max_threads", "svt_set_max_threads", "svt_dev_aperm_ws_bytes", "svt_dev_aperm", "svt_dev_aperm_route_counts", "svt_aperm_SVT", "svt_transpose_2D_SVT", "svt_dev_transpose_ws_bytes", "svt_dev_transpose", "svt_dev_colstats", "svt_dev_rowstats_ws_bytes", "svt_dev_rowsums", "svt_dev_rowsum",
]


class HipBackendError(RuntimeError):
    pass


_lib = None
_ready = False


def load_library() -> ctypes.CDLL:
    """dlopen only -- no GPU needed (used by the ABI tests on CPU boxes)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipBackendError(
                f"{LIB_PATH} is missing: build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  sparsearray_amd has no CPU path.")
        # torch ships a HIP runtime of its own; when libsvt_hip.so brings in the system one first, torch's
        # later initialisation finds "no HIP GPUs".  Device memory and streams come from torch
        # (sparsearray_amd/device.py), so load it first and let both use one runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.svt_last_error.restype = ctypes.c_char_p
        _lib.svt_device_arch.restype = ctypes.c_char_p
        _lib.svt_init.argtypes = [ctypes.c_int]
        _lib.svt_init.restype = ctypes.c_int
    return _lib


def init(device: int | None = None) -> ctypes.CDLL:
    """Load the library and bind this process to one GPU (LOCAL_RANK by default)."""
    global _ready
    lib = load_library()
    if not _ready:
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        if lib.svt_init(device) != 0:
            raise HipBackendError(lib.svt_last_error().decode())
        _ready = True
    return lib


def hip_dispatcher():
    from ._dispatch import CAbiDispatcher
    return CAbiDispatcher(init(), "svt_")
