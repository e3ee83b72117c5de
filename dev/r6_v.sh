#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python dev/r6_case_a.py 2>&1 | grep "n_iter" | cut -c1-460
timeout 900 python dev/fuzz_rank.py 2>&1 | grep -v amdgpu | tail -3
timeout 900 python dev/fuzz_clip.py 2>&1 | grep -v amdgpu | tail -2
FUZZ6_ONLY=off FUZZ_GEMM=fp32 timeout 1500 python dev/fuzz_round6.py 72 20 2>&1 | grep "FAIL\|failures" | cut -c1-260
timeout 3000 python -m pytest tests/ -x -q -m gpu > gpurun_out/r6_v_gpu_suite.txt 2>&1; grep -n "passed\|failed" gpurun_out/r6_v_gpu_suite.txt | tail -3
