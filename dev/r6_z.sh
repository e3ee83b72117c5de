#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export PETAL_DEBUG_ATTEMPTS=1
timeout 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/r6_z_gpu_suite.txt 2>&1; grep -n "passed\|failed" gpurun_out/r6_z_gpu_suite.txt | tail -3
grep -n "petal rank-offset" gpurun_out/r6_z_gpu_suite.txt | tail -40 | cut -c1-260
