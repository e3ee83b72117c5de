#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python dev/fuzz_all.py 11 60 2>&1 | grep -v amdgpu | grep -c "^ok" 
timeout 900 python dev/fuzz_all.py 11 60 2>&1 | grep "FAIL" | head -20
FUZZ_GEMM=fp32 timeout 600 python dev/fuzz_all.py 12 30 2>&1 | grep "FAIL" | head -20
timeout 600 python dev/fuzz_rpca.py 13 60 2>&1 | grep -v amdgpu | tail -4
timeout 600 python dev/fuzz_rank.py 2>&1 | grep -v amdgpu | tail -4
timeout 600 python dev/fuzz_clip.py 2>&1 | grep -v amdgpu | tail -3
echo sweep-done
