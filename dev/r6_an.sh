#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python dev/api_sweep.py 2>&1 | grep -v amdgpu | tail -4 | cut -c1-200
timeout 600 python dev/eigh_cluster_big.py 2>&1 | grep -v amdgpu | tail -3 | cut -c1-200
timeout 600 python dev/fuzz_accurate.py 2>&1 | grep -v amdgpu | tail -3 | cut -c1-200
timeout 900 python dev/fuzz_round5.py 5 30 2>&1 | grep "FAIL\|failures" | cut -c1-200
