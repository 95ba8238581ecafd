"""GPU: the C++ host layer (old-kaldi-git_amd/host/kaldi-hip.h) — a C++ program written
like the reference's cu-matrix-test.cc, linked against libkaldi_hip.so."""
import os
import subprocess

import pytest

from conftest import ROOT, pkg

pytestmark = pytest.mark.gpu


def test_cpp_host_api_program(api):
    exe = pkg("build").build_host_test()
    env = dict(os.environ)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all tests passed" in out.stdout
