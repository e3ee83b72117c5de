#!/bin/bash
# usage: dev/build_var.sh NAME [-DFLAG ...]  -> dev/libpetal_NAME.so (an A/B variant of the library; select it with PETAL_HIP_LIBRARY)
name=$1; shift
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o dev/libpetal_$name.so \
  petal-decomposition_amd/csrc/hip_ops.hip petal-decomposition_amd/csrc/algo.cpp petal-decomposition_amd/csrc/api.cpp petal-decomposition_amd/csrc/rccl.cpp -ldl 2>&1 | grep -E "error" ; ls -la dev/libpetal_$name.so
