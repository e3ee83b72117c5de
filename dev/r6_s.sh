#!/bin/bash
cd "$GRAFT_REPO_ROOT"
FUZZ6_ONLY=off timeout 1500 python dev/fuzz_round6.py 71 40 > gpurun_out/r6_s_off.txt 2>&1; grep "FAIL\|failures" gpurun_out/r6_s_off.txt | cut -c1-260
FUZZ6_ONLY=off FUZZ_GEMM=fp32 timeout 1500 python dev/fuzz_round6.py 72 20 > gpurun_out/r6_s_off_fp32.txt 2>&1; grep "FAIL\|failures" gpurun_out/r6_s_off_fp32.txt | cut -c1-260
