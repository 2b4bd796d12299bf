"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the SVT hot path.

Nothing under ``sparsearray_amd/`` imports this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do.
"""
from .oracle import load_oracle, oracle_session  # noqa: F401
