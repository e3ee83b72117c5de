"""The C-ABI library builds, loads on a CPU-only box and exports every symbol include/petal_hip.h declares
(no compute calls: there is no GPU here), and it refuses to create a context without a gfx950 device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "petal_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(petal_[a-z0-9_]+)\s*\(", text))
    names -= {"petal_allreduce_fn"}
    return sorted(names)


@pytest.fixture(scope="module")
def lib_path():
    import importlib.util
    spec = importlib.util.spec_from_file_location("petal_build", os.path.join(ROOT, "petal-decomposition_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build()


def test_header_symbols_exported(lib_path):
    lib = ctypes.CDLL(lib_path)
    names = _declared()
    assert len(names) == 25, names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/petal_hip.h but not exported"


def test_python_binding_covers_header(lib_path):
    import petal_decomposition_amd as petal
    assert sorted(n for n, _, _ in petal.ABI) == _declared()
    petal.load_library(lib_path, preload_torch=False)


def test_no_cpu_fallback(lib_path):
    import petal_decomposition_amd as petal
    import shutil
    if shutil.which("rocminfo") and os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    lib = petal.load_library(lib_path, preload_torch=False)
    with pytest.raises(petal.DeviceError):
        petal.Context(0, lib=lib)


def test_product_does_not_reference_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "petal-decomposition_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(import|from)\s+oracle", src, flags=re.M), f
                assert not re.search(r"#\s*include\s*[\"<][^\n]*oracle", src), f
