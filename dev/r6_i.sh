#!/bin/bash
cd "$GRAFT_REPO_ROOT"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o dev/eig_stress dev/eig_stress.hip petal-decomposition_amd/csrc/algo.cpp petal-decomposition_amd/csrc/api.cpp petal-decomposition_amd/csrc/rccl.cpp -ldl > gpurun_out/r6_i_build.txt 2>&1; tail -3 gpurun_out/r6_i_build.txt
./dev/eig_stress > gpurun_out/r6_i_eig_stress.txt 2>&1; grep -c " ok" gpurun_out/r6_i_eig_stress.txt; grep -i "FAIL\|budget" gpurun_out/r6_i_eig_stress.txt | head
