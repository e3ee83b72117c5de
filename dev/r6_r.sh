#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 3000 python -m pytest tests/ -x -q -m gpu > gpurun_out/r6_r_gpu_suite.txt 2>&1; grep -n "passed\|failed" gpurun_out/r6_r_gpu_suite.txt | tail -3
