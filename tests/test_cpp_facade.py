"""Builds tests/cpp/facade_tests.cpp (the reference's unit tests against the C++ facade
include/petal_decomposition.hpp) and runs it: against the host simulation on CPU, against libpetal_hip.so on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "facade_tests.cpp")


def _build_and_run(lib_path, tag):
    out = os.path.join(ROOT, "tests", "_build", f"facade_tests_{tag}")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    libdir, libname = os.path.split(lib_path)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), SRC, "-o", out,
                           "-L", libdir, f"-l:{libname}", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    res = subprocess.run([out], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "all reference unit tests passed" in res.stdout


def test_facade_on_host_simulation():
    import hostsim
    _build_and_run(hostsim.build(), "hostsim")


@pytest.mark.gpu
def test_facade_on_gpu():
    lib = os.path.join(ROOT, "petal-decomposition_amd", "libpetal_hip.so")
    assert os.path.exists(lib), "libpetal_hip.so missing: run python __graft_entry__.py build"
    _build_and_run(lib, "hip")
