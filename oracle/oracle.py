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


def oracle_dispatcher():
    from sparsearray_amd._dispatch import CAbiDispatcher
    return CAbiDispatcher(load_oracle(), "orc_")


def oracle_session():
    from sparsearray_amd.api import Session
    return Session(oracle_dispatcher())
