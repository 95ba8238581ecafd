"""CPU: the oracle (restatement) against the golden vectors generated from the
reference's own compiled CPU code (tests/golden/make_golden.py) — this is what
pins the oracle for SURVEY.md §8 rows a1-a9."""
import pytest

import cases


@pytest.mark.parametrize("case", cases.ALL_GOLDEN, ids=lambda f: f.__name__)
def test_oracle_matches_reference_golden(oracle, case):
    case(oracle)
