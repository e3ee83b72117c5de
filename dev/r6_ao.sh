#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q -m gpu -k "rpca or rebased or close_eigen or three_iter or low_iter or sharded or rank or clip or golden" > gpurun_out/r6_ao.txt 2>&1; grep -n "passed\|failed\|Error" gpurun_out/r6_ao.txt | head -5
timeout 600 python dev/fuzz_rank.py 2>&1 | grep -v amdgpu | tail -2
timeout 600 python dev/fuzz_clip.py 2>&1 | grep -v amdgpu | tail -2
timeout 600 python dev/fuzz_rpca.py 13 60 2>&1 | grep -v amdgpu | tail -2
FUZZ6_ONLY=f64 timeout 900 python dev/fuzz_round6.py 91 60 2>&1 | grep "FAIL\|failures" | cut -c1-220
