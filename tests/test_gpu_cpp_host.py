"""GPU: the C++ host layer (old-kaldi-git_amd/host/kaldi-hip.h) — a C++ program written
like the reference's cu-matrix-test.cc, linked against libkaldi_hip.so."""
import os
import subprocess

import pytest

from conftest import ROOT, pkg

pytestmark = pytest.mark.gpu


def test_cpp_host_api_program(api, tmp_path):
    exe = pkg("build").build_host_test()
    env = dict(os.environ)
    # expectation for the C++ feature classes: the Python mirror on the same waveform
    import numpy as np
    import torch
    rng = np.random.default_rng(3)
    wave = (2000 * np.sin(np.arange(16000) * 0.05) + 300 * rng.standard_normal(16000)).astype(np.float32)
    mf = api.Mfcc(num_bins=40, num_ceps=40, low_freq=40.0, high_freq=-200.0)
    x = mf.compute(torch.from_numpy(wave).cuda())
    y = api.apply_cmvn(api.acc_cmvn_stats(x), True, x.clone())
    d = api.compute_deltas(y, 2, 2)
    path = str(tmp_path / "features.bin")
    with open(path, "wb") as f:
        f.write(np.asarray([len(wave), x.shape[0], x.shape[1]], np.int32).tobytes())
        f.write(wave.tobytes())
        f.write(np.ascontiguousarray(x.cpu().numpy()).tobytes())
        f.write(np.ascontiguousarray(d.cpu().numpy()).tobytes())
    out = subprocess.run([exe, path], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all tests passed" in out.stdout


def test_cpp_reference_cu_matrix_unit_tests():
    """tests/cpp/cu_matrix_test.cc: the CuMatrix<Real> / CuVector<Real> primitives through the reference's class API
    (call syntax of cudamatrix/cu-matrix.h), for float and double, against identities / closed forms / host loops
    (this repo's own test designs), views, copies between precisions, and LatticeFasterDecoder(fst, config).Decode(&decodable)."""
    exe = pkg("build").build_host_test("cu_matrix_test")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all tests passed" in out.stdout
