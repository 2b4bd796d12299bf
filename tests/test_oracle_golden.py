"""The oracle (CPU restatement of the reference) against every golden vector
the reference's own tests hold for the hot path."""
import pytest

from helpers import check_case, golden_cases

CASES = golden_cases()


@pytest.mark.parametrize("lacunar", [True, False], ids=["lacunar", "plain"])
@pytest.mark.parametrize("case", CASES, ids=[f"{c['id']}-{c['fn']}" for c in CASES])
def test_oracle_matches_reference_vectors(oracle, case, lacunar):
    check_case(oracle, case, lacunar=lacunar)
