"""Builds and runs tests/cpp/reference_suite.cpp: the reference's NUnit tests for the path
restated against the C++ host layer include/SdfKit.hpp (the reference is compiled code; .NET
is absent, C++ is the host language available)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp):
    from sdfkit_amd import _native as N
    N.lib()  # makes sure libsdfkit_hip.so exists
    exe = os.path.join(tmp, "reference_suite")
    libdir = os.path.join(ROOT, "sdfkit_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "reference_suite.cpp"), "-o", exe,
           "-L", libdir, "-lsdfkit_hip", f"-Wl,-rpath,{libdir}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_cpp_host_layer_compiles(tmp_path):
    """CPU-side: the header-only mirror compiles and links against the C ABI."""
    assert os.path.exists(_build(str(tmp_path)))


@pytest.mark.gpu
def test_reference_suite_through_cpp_host_layer(tmp_path, gpu):
    exe = _build(str(tmp_path))
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(p.stdout[-3000:], p.stderr[-2000:])
    assert p.returncode == 0, p.stdout[-3000:]
    assert "22 tests, 0 failures" in p.stdout
