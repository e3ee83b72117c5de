#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python dev/fuzz_round6.py 61 40 > gpurun_out/r6_n_fuzz6.txt 2>&1; grep -E "FAIL|failures" gpurun_out/r6_n_fuzz6.txt | cut -c1-260
FUZZ_GEMM=fp32 timeout 1200 python dev/fuzz_round6.py 62 25 > gpurun_out/r6_n_fuzz6_fp32.txt 2>&1; grep -E "FAIL|failures" gpurun_out/r6_n_fuzz6_fp32.txt | cut -c1-260
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not_row_major" 2>&1 | tail -2
