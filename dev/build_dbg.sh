#!/bin/bash
# builds dev/libpetal_dbg.so (the library with -DPETAL_DEBUG_COUNTERS: in-kernel phase counters)
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DPETAL_DEBUG_COUNTERS -o dev/libpetal_dbg.so \
  petal-decomposition_amd/csrc/hip_ops.hip petal-decomposition_amd/csrc/algo.cpp petal-decomposition_amd/csrc/api.cpp petal-decomposition_amd/csrc/rccl.cpp -ldl
