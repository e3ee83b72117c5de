"""AddressSanitizer + UndefinedBehaviorSanitizer over the PRODUCT's host code (api.cpp, algo.cpp, rccl.cpp) driven by the
reference's unit tests through the C++ facade, linked against the host simulation of the device ops.  CPU only (GPU
sanitizers are not available on the test pool)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_is_clean_under_asan_ubsan():
    out = os.path.join(ROOT, "tests", "_build", "facade_asan")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    srcs = [os.path.join(ROOT, "tests", "cpp", "facade_tests.cpp"),
            os.path.join(ROOT, "petal-decomposition_amd", "csrc", "api.cpp"),
            os.path.join(ROOT, "petal-decomposition_amd", "csrc", "algo.cpp"),
            os.path.join(ROOT, "petal-decomposition_amd", "csrc", "rccl.cpp"),
            os.path.join(ROOT, "oracle", "cpu_ops.cpp")]
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-I", os.path.join(ROOT, "include")] + srcs + ["-ldl", "-o", out]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("this g++ has no sanitizer runtime")
    assert build.returncode == 0, build.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    res = subprocess.run([out], capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, (res.stdout + res.stderr)[-4000:]
    assert "all reference unit tests passed" in res.stdout
    assert "ERROR: AddressSanitizer" not in res.stderr and "runtime error" not in res.stderr
