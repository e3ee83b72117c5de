"""Builds tests/_build/libpetal_hostsim.so: the PRODUCT's host algorithms (api.cpp, algo.cpp) linked
against oracle/cpu_ops.cpp, the host-memory simulation of the device-op layer.  Test infrastructure:
lets the CPU suite exercise the host logic, the C ABI error contract and the sharded collective path
without a GPU.  The product library never contains this code."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "_build", "libpetal_hostsim.so")
SRCS = [os.path.join(ROOT, "petal-decomposition_amd", "csrc", "api.cpp"),
        os.path.join(ROOT, "petal-decomposition_amd", "csrc", "algo.cpp"),
        os.path.join(ROOT, "petal-decomposition_amd", "csrc", "rccl.cpp"),
        os.path.join(ROOT, "oracle", "cpu_ops.cpp")]
HDRS = [os.path.join(ROOT, "petal-decomposition_amd", "csrc", h) for h in ("ops.h", "ctx.h")] + \
       [os.path.join(ROOT, "include", "petal_hip.h")]


def build() -> str:
    if os.environ.get("PETAL_HOSTSIM_LIBRARY"):   # (a sanitizer build of the same sources: dev/asan_hostsim.sh)
        return os.environ["PETAL_HOSTSIM_LIBRARY"]
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    newest = max(os.path.getmtime(p) for p in SRCS + HDRS)
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < newest:
        tmp = OUT + f".{os.getpid()}.tmp"
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", tmp] + SRCS + ["-ldl"])
        os.replace(tmp, OUT)
    return OUT


def context():
    import petal_decomposition_amd as petal
    lib = petal.load_library(build(), preload_torch=False)
    lib._petal_host_buffers = True   # this library's "device" pointers are host pointers (collective hook)
    return petal.Context(0, lib=lib)
