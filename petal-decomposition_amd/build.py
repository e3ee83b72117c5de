"""Builds libpetal_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libpetal_hip.so")
SOURCES = ["hip_ops.hip", "algo.cpp", "api.cpp", "rccl.cpp"]
HEADERS = ["ops.h", "ctx.h", os.path.join("..", "..", "include", "petal_hip.h")] + \
    [os.path.join("kernels", f) for f in sorted(os.listdir(os.path.join(CSRC, "kernels"))) if f.endswith(".inc")]   # the parts of hip_ops.hip


def build(force: bool = False, verbose: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS]
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(p) for p in deps):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tmp = OUT + f".{os.getpid()}.tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", tmp] + srcs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
