#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 3400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r6_y_gpu_suite.txt 2>&1; grep -n "passed\|failed" gpurun_out/r6_y_gpu_suite.txt | tail -3
