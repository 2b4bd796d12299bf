"""Parity tests proper: the HIP path, called through the C ABI, against the
golden vectors of the reference's own tests.  Integer / logical results must
be bit-exact; double results within 1e-6 relative with matching NaN class (and
matching NA class where the reference returns NA_real_ explicitly)."""
import pytest

from helpers import check_case, golden_cases

pytestmark = pytest.mark.gpu
CASES = golden_cases()


@pytest.mark.parametrize("lacunar", [True, False], ids=["lacunar", "plain"])
@pytest.mark.parametrize("case", CASES, ids=[f"{c['id']}-{c['fn']}" for c in CASES])
def test_hip_matches_reference_vectors(hip, case, lacunar):
    check_case(hip, case, lacunar=lacunar, gpu=True)
