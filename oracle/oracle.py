"""ctypes binding of oracle/libsvt_oracle.so (TEST INFRASTRUCTURE ONLY).

``oracle_session()`` returns a ``sparsearray_amd.api.Session`` whose ``.Call``
dispatcher is the CPU restatement of the reference, so the parity tests push
the oracle and the HIP path through the very same R-level logic.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libsvt_oracle.so")
_lib = None


def build_oracle(force: bool = False) -> str:
    src = os.path.join(_HERE, "svt_oracle.c")
    stale = (not os.path.exists(_LIB) or
             os.path.getmtime(_LIB) < os.path.getmtime(src))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "libsvt_oracle.so"])
    return _LIB


def load_oracle() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build_oracle()
        _lib = ctypes.CDLL(_LIB)
        _lib.orc_set_max_threads.argtypes = [ctypes.c_int]
        _lib.orc_set_max_threads.restype = ctypes.c_int
        _lib.orc_get_max_threads.restype = ctypes.c_int
        _lib.orc_get_num_procs.restype = ctypes.c_int
    return _lib


def _positive_padded_median(x, padding):
    """.positive_padded_median(), R/SparseArray-matrixStats.R:695-708 (x >= 0, padding < len(x))."""
    import numpy as np
    n = len(x) + padding
    xs = np.sort(x)                                  # sort(x, partial=k)[k] picks the same elements
    if n % 2 == 1:
        partial = (n + 1) // 2 - padding
        return float(xs[partial - 1])
    i1 = n // 2 - padding
    return float(np.mean(xs[i1 - 1:i1 + 1]))


def _padded_median(x, padding, na_rm):
    """.padded_median(), R/SparseArray-matrixStats.R:712-758: median(c(x, integer(padding)))."""
    import numpy as np
    from sparsearray_amd import NA_real
    x = np.asarray(x, dtype=np.float64)
    if na_rm:
        x = x[~np.isnan(x)]
    elif np.isnan(x).any():
        return NA_real
    n = len(x) + padding
    if n == 0:
        return NA_real
    if padding > len(x):
        return 0.0
    pos = x > 0
    pos_count = int(pos.sum())
    nonpos_count = n - pos_count
    if pos_count > nonpos_count:
        return _positive_padded_median(x[pos], nonpos_count)
    neg_count = len(x) - pos_count
    nonneg_count = n - neg_count
    if neg_count > nonneg_count:
        return -_positive_padded_median(-x[~pos], nonneg_count)
    if n % 2 == 1:
        return 0.0
    half = n // 2
    right = float(x[pos].min()) if pos_count == half else 0.0
    left = float(x[~pos].max()) if neg_count == half else 0.0
    return (left + right) * 0.5


def oracle_dispatcher():
    import numpy as np
    from sparsearray_amd import NA_integer
    from sparsearray_amd._dispatch import CAbiDispatcher

    class OracleDispatcher(CAbiDispatcher):
        # .colMedians_SVT_SparseMatrix, R/SparseArray-matrixStats.R:761-784: pure R in the
        # reference, restated leaf by leaf in Python
        def C_colMedians_SVT(self, x, na_rm):
            nrow, ncol = x.dim
            ans = np.zeros(ncol)
            for j, lf in enumerate(x.leaves):
                if lf is None:
                    continue
                vals = lf[1]
                if vals is None:                     # lacunar leaf: all ones
                    vals = np.ones(len(lf[0]))
                vals = np.asarray(vals)
                if vals.dtype != np.float64:
                    v = vals.astype(np.float64)
                    v[vals == NA_integer] = np.nan
                    vals = v
                ans[j] = _padded_median(vals, nrow - len(vals), bool(na_rm))
            return ans

    return OracleDispatcher(load_oracle(), "orc_")


def oracle_session():
    from sparsearray_amd.api import Session
    return Session(oracle_dispatcher())
