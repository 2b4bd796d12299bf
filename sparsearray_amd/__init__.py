"""sparsearray_amd -- MI355X-native backend for SparseArray's SVT hot path.

Host-side mirror of the reference's operator interface for crossprod / %*%,
col*/row* matrixStats, summarization and rowsum/colsum, over the C-ABI HIP
library ``libsvt_hip.so`` (include/svt_hip.h).  There is no CPU compute path
in this package: using any operator without the HIP library raises.
"""
from .svt import (NA_integer, NA_logical, NA_real, SVT_SparseArray,  # noqa: F401
                  is_NA_real, is_NaN_real)
from .api import Session, SparseArrayError, SparseArrayUnsupported  # noqa: F401


def hip_session():
    """The product session: every entry point runs on the GPU."""
    from ._hip import hip_dispatcher
    return Session(hip_dispatcher())
