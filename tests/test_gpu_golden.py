"""GPU: the HIP path (through the C-ABI) against the reference's golden vectors."""
import pytest

import cases
from gpu_impl import GpuImpl

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", cases.ALL_GOLDEN, ids=lambda f: f.__name__)
def test_hip_matches_reference_golden(api, case):
    case(GpuImpl(api))
