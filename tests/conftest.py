import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "old-kaldi-git_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(sub=None):
    return importlib.import_module(PKG + ("." + sub if sub else ""))


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure)."""
    from oracle import binding
    if not os.path.exists(binding.ORACLE_SO):
        binding.build(ref=os.path.exists("/root/reference/src"))
    return binding.OracleLib("ko")


@pytest.fixture(scope="session")
def ref_lib():
    """The reference's own CPU code compiled from /root/reference (only where present)."""
    from oracle import binding
    if not binding.have_ref():
        if os.path.exists("/root/reference/src"):
            binding.build(ref=True)
        else:
            pytest.skip("oracle/_ref not built and /root/reference absent")
    return binding.OracleLib("ref")


@pytest.fixture(scope="session")
def api():
    """The product's host API; requires the HIP library and a GPU."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    a = pkg("api")
    a.select_gpu(0)
    return a


@pytest.fixture
def rng():
    return np.random.default_rng(1234)


GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))
