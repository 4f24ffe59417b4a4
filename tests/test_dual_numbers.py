"""csrc/gp_dual.h -- the forward-mode type a run-time compiled covariance function is instantiated with for
gphip_loglik_grad -- compiled for the HOST with g++ (tests/dual/dual_check.cpp) and checked against central differences:
every operator, comparison and math function the header differentiates, mixed scalar / dual arithmetic, constant exponents on
negative bases, the P(k) accessor.  (The device side of the same text: tests/test_custom_kernel.py compiles it under hiprtc,
tests/test_gpu_custom_kernel.py checks the gradients it produces against the oracle.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dual_numbers_match_central_differences(tmp_path):
    cxx = shutil.which("g++") or shutil.which("c++")
    if not cxx:
        pytest.skip("no host C++ compiler")
    exe = tmp_path / "dual_check"
    res = subprocess.run([cxx, "-O1", "-std=c++17", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Werror", "-I" + os.path.join(ROOT, "bayesianinference_amd", "csrc"),
                          "-o", str(exe), os.path.join(ROOT, "tests", "dual", "dual_check.cpp")], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    rows = [ln.split() for ln in out.stdout.strip().splitlines()]
    assert len(rows) == 13, out.stdout
    for name, verr, gerr in rows:
        assert float(verr) <= 1e-14, (name, verr)
        assert float(gerr) <= 2e-8, (name, gerr)          # (central differences with h = 1e-6 |theta|: ~1e-10 .. 1e-9)
