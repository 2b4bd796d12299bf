"""Short runs of the differential fuzzers of tools/debug (random shapes, skewed columns, forced
row-split counts, every launch path of the statistics kernels) as part of the GPU suite."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,ncases,seed", [("fuzz_pbc.py", 60, 101), ("fuzz_stats.py", 24, 102), ("fuzz_transpose.py", 40, 103), ("fuzz_spmm.py", 40, 104), ("fuzz_gram.py", 40, 105)])
def test_fuzzers_find_nothing(hip, script, ncases, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "debug", script), str(ncases), str(seed)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "MISMATCH" not in r.stdout
    assert "Memory access fault" not in r.stdout + r.stderr
