#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "close_eigenvalues or three_iterations or rccl or verdict" 2>&1 | tail -2
bash dev/fuzz_sweep.sh > gpurun_out/r6_u_fuzz_sweep.txt 2>&1; tail -30 gpurun_out/r6_u_fuzz_sweep.txt | cut -c1-220
