import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_session
    from oracle.oracle import load_oracle
    # The oracle's OpenMP loops take every core they see; on a shared GPU box the process owns
    # a share of them (16 for one GPU) and oversubscription makes the large cases crawl.
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    load_oracle().orc_set_max_threads(max(1, min(ncpu, 16)))
    return oracle_session()


@pytest.fixture(scope="session")
def hip():
    """The product session.  Fails loudly when libsvt_hip.so is missing or no
    GPU is visible -- there is no fallback."""
    import sparsearray_amd
    return sparsearray_amd.hip_session()
