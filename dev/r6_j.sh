#!/bin/bash
cd "$GRAFT_REPO_ROOT"
bash dev/fuzz_sweep.sh > gpurun_out/r6_j_fuzz_sweep.txt 2>&1; tail -30 gpurun_out/r6_j_fuzz_sweep.txt | cut -c1-220
timeout 900 python dev/soak.py 1500 > gpurun_out/r6_j_soak.txt 2>&1; tail -4 gpurun_out/r6_j_soak.txt
